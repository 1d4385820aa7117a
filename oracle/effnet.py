"""CPU oracle: timm-0.4.12 ``efficientnet_b3a`` forward (TEST INFRASTRUCTURE — oracle/__init__.py).

PARITY UNPINNED: timm==0.4.12 (reference requirements.txt:164) is not vendored/installed and the
reference has no test for the backbone.  Structure restated from timm 0.4.12's published
``efficientnet.py`` / ``efficientnet_blocks.py`` (decode of the efficientnet_b0 arch strings with
channel_multiplier 1.2, depth_multiplier 1.4, stem 32, head 1280, SE ratio 0.25 of the block INPUT
channels, symmetric k//2 padding, BN eps 1e-5, SiLU) and anchored by the exact parameter count
12 233 232 (10 696 232 without the 1000-way classifier), see tests/test_oracle_backbones.py.

Call sites it stands in for: ``timm.create_model('efficientnet_b3a')`` at
inference/inference.py:102,110 ; ``model(x)`` :199-201 ; ``forward_features`` / ``classifier``
at train/train_efficientnet.py:226,230 ; ``get_fm`` (AvgPool2d) train/train.py:84-103.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .common import Rounder, SeededInit, bn_of, fold_bn, make_divisible

BN_EPS = 1e-5
# (type, kernel, stride, expand, out_channels_base, repeats_base) — efficientnet_b0 arch_def
_B0 = [("ds", 3, 1, 1, 16, 1), ("ir", 3, 2, 6, 24, 2), ("ir", 5, 2, 6, 40, 2), ("ir", 3, 2, 6, 80, 3),
       ("ir", 5, 1, 6, 112, 3), ("ir", 5, 2, 6, 192, 4), ("ir", 3, 1, 6, 320, 1)]


def arch(channel_multiplier=1.2, depth_multiplier=1.4):
    """-> dict(stem, stages=[[block dicts]], head).  Block: type,k,s,cin,cout,mid,se_rd."""
    rc = lambda c: make_divisible(c * channel_multiplier, 8)
    stem = rc(32)
    stages, cin = [], stem
    for (typ, k, s, e, c, r) in _B0:
        cout = rc(c)
        blocks = []
        for b in range(int(math.ceil(r * depth_multiplier))):
            blocks.append(dict(type=typ, k=k, s=s if b == 0 else 1, cin=cin, cout=cout,
                               mid=cin * e, se_rd=make_divisible(cin * 0.25, 1)))
            cin = cout
        stages.append(blocks)
    return dict(stem=stem, stages=stages, head=rc(1280))


def init_state_dict(seed: int, num_classes: int = 1000):
    a = arch()
    g = SeededInit(seed)
    g.conv("conv_stem.weight", (a["stem"], 3, 3, 3))
    g.bn("bn1", a["stem"])
    for si, st in enumerate(a["stages"]):
        for bi, b in enumerate(st):
            p = f"blocks.{si}.{bi}"
            mid, k = b["mid"], b["k"]
            if b["type"] == "ir":
                g.conv(f"{p}.conv_pw.weight", (mid, b["cin"], 1, 1))
                g.bn(f"{p}.bn1", mid)
                g.conv(f"{p}.conv_dw.weight", (mid, 1, k, k))
                g.bn(f"{p}.bn2", mid)
            else:
                g.conv(f"{p}.conv_dw.weight", (mid, 1, k, k))
                g.bn(f"{p}.bn1", mid)
            g.conv(f"{p}.se.conv_reduce.weight", (b["se_rd"], mid, 1, 1))
            g.vec(f"{p}.se.conv_reduce.bias", b["se_rd"], "normal", 0.0, 0.1)
            g.conv(f"{p}.se.conv_expand.weight", (mid, b["se_rd"], 1, 1))
            g.vec(f"{p}.se.conv_expand.bias", mid, "normal", 0.0, 0.1)
            if b["type"] == "ir":
                # linear (no activation) projection: gain 1 keeps the residual stream bounded
                g.conv(f"{p}.conv_pwl.weight", (b["cout"], mid, 1, 1), gain=1.0)
                g.bn(f"{p}.bn3", b["cout"])
            else:
                g.conv(f"{p}.conv_pw.weight", (b["cout"], mid, 1, 1), gain=1.0)
                g.bn(f"{p}.bn2", b["cout"])
    g.conv("conv_head.weight", (a["head"], a["stages"][-1][-1]["cout"], 1, 1))
    g.bn("bn2", a["head"])
    if num_classes > 0:
        g.conv("classifier.weight", (num_classes, a["head"]), gain=1.0)
        g.vec("classifier.bias", num_classes, "normal", 0.0, 0.1)
    return g.sd


def _conv_bn(x, sd, conv, bn, rb, stride=1, pad=0, groups=1, act=True):
    w, b = fold_bn(sd[f"{conv}.weight"], bn_of(sd, bn), BN_EPS)
    y = F.conv2d(x, rb(w), b, stride=stride, padding=pad, groups=groups)
    return F.silu(y) if act else y


def _se(x, sd, p, rb):
    """SE on the fp32 (pre-rounding) activation: mean -> conv_reduce(+b) -> SiLU -> conv_expand(+b)
    -> sigmoid.  Returns the gate.  The HIP path stores the two FC matrices in bf16 (``rb`` in sim mode; every image's
    workgroup streams them from L2) and keeps biases, accumulation and the gate in fp32."""
    s = x.mean((2, 3), keepdim=True)
    r = F.silu(F.conv2d(s, rb(sd[f"{p}.se.conv_reduce.weight"]), sd[f"{p}.se.conv_reduce.bias"]))
    return torch.sigmoid(F.conv2d(r, rb(sd[f"{p}.se.conv_expand.weight"]), sd[f"{p}.se.conv_expand.bias"]))


def forward_features(sd, x, sim_bf16=False, taps=None):
    """(B,3,H,W) fp32 -> un-pooled (B,1536,H/32,W/32) fp32, = timm ``forward_features``.

    ``sim_bf16`` reproduces the HIP path's rounding points: folded conv weights and every stored
    activation are rounded to bf16, accumulation/BN-bias/activation functions/SE stay fp32, the SE
    squeeze averages the un-rounded depthwise output, the gated tensor is re-rounded before conv_pwl.
    ``taps`` (dict) collects intermediate tensors by name for per-layer parity tests."""
    rb = Rounder(sim_bf16)
    a = arch()
    x = rb(_conv_bn(x, sd, "conv_stem", "bn1", rb, stride=2, pad=1))
    if taps is not None:
        taps["stem"] = x
    for si, st in enumerate(a["stages"]):
        for bi, b in enumerate(st):
            p = f"blocks.{si}.{bi}"
            k, s, mid = b["k"], b["s"], b["mid"]
            sc = x
            if b["type"] == "ir":
                x = rb(_conv_bn(x, sd, f"{p}.conv_pw", f"{p}.bn1", rb))
                d = _conv_bn(x, sd, f"{p}.conv_dw", f"{p}.bn2", rb, stride=s, pad=k // 2, groups=mid)
                pw, bnl = f"{p}.conv_pwl", f"{p}.bn3"
            else:
                d = _conv_bn(x, sd, f"{p}.conv_dw", f"{p}.bn1", rb, stride=s, pad=k // 2, groups=mid)
                pw, bnl = f"{p}.conv_pw", f"{p}.bn2"
            gate = _se(d, sd, p, rb)
            x = rb(rb(d) * gate)
            x = _conv_bn(x, sd, pw, bnl, rb, act=False)
            if b["s"] == 1 and b["cin"] == b["cout"]:
                x = x + sc
            x = rb(x)
            if taps is not None:
                taps[p] = x
    x = rb(_conv_bn(x, sd, "conv_head", "bn2", rb))
    return x


def pool(fm):
    """get_fm, train/train.py:101-103: AvgPool2d((H,W)) + reshape -> (B,C)."""
    return torch.reshape(F.avg_pool2d(fm, (fm.shape[2], fm.shape[3])), (-1, fm.shape[1]))


def forward(sd, x, sim_bf16=False):
    """= timm ``forward``: classifier(global_pool(forward_features(x))); identity classifier when
    the state dict has none (num_classes=0)."""
    rb = Rounder(sim_bf16)
    f = pool(forward_features(sd, x, sim_bf16))
    if "classifier.weight" in sd:
        return F.linear(rb(f), rb(sd["classifier.weight"]), sd["classifier.bias"])
    return f
