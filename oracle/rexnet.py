"""CPU oracle: timm-0.4.12 ``rexnet_{100,130,150,200}`` forward (TEST INFRASTRUCTURE — oracle/__init__.py).

PARITY UNPINNED (timm not vendored/installed; the reference holds no fixture).  Structure restated from timm
0.4.12's published ``rexnet.py`` (``_block_cfg``: layers [1,2,2,3,3,5], strides [1,2,2,2,1,2], expansion 6 after
the first block, channels growing linearly 16 -> 16+180 scaled by width_mult with ch_div=1, SEWithNorm ratio 1/12
from the third stage on, swish after conv_exp, ReLU6 after the depthwise/SE, partial-channel shortcut
``x[:, :in_chs] += shortcut``) and anchored by timm's published parameter counts (results table: rexnet_100 4.80 M,
rexnet_130 7.56 M, rexnet_150 9.73 M, rexnet_200 16.37 M; this restatement gives 4 796 873 / 7 557 091 /
9 728 593 / 16 366 620), see tests/test_oracle_backbones.py.

Call sites: ``timm.create_model('rexnet_150')`` inference/inference.py:268 (default), ``forward_features`` +
``head`` train/train.py:194-195.
"""
from __future__ import annotations

from math import ceil

import torch
import torch.nn.functional as F

from .common import Rounder, SeededInit, bn_of, fold_bn, make_divisible

BN_EPS = 1e-5


def block_cfg(width_mult=1.0, depth_mult=1.0, initial_chs=16, final_chs=180, se_ratio=1 / 12.0, ch_div=1):
    layers = [1, 2, 2, 3, 3, 5]
    strides = [1, 2, 2, 2, 1, 2]
    layers = [ceil(e * depth_mult) for e in layers]
    strides = sum([[e] + [1] * (layers[i] - 1) for i, e in enumerate(strides)], [])
    exp_ratios = [1] * layers[0] + [6] * sum(layers[1:])
    depth = sum(layers) * 3
    base_chs = initial_chs / width_mult if width_mult < 1.0 else initial_chs
    out = []
    for _ in range(depth // 3):
        out.append(make_divisible(round(base_chs * width_mult), divisor=ch_div))
        base_chs += final_chs / (depth // 3 * 1.0)
    se = [0.0] * (layers[0] + layers[1]) + [se_ratio] * sum(layers[2:])
    return list(zip(out, exp_ratios, strides, se))


def arch(width_mult):
    stem = make_divisible(round(32 * width_mult), divisor=1)
    blocks, prev = [], stem
    for chs, e, s, se in block_cfg(width_mult):
        dw = make_divisible(round(prev * e), divisor=1) if e != 1 else prev
        rd = make_divisible(int(dw * se), divisor=1) if se > 0 else 0
        blocks.append(dict(cin=prev, cout=chs, e=e, s=s, dw=dw, rd=rd))
        prev = chs
    return dict(stem=stem, blocks=blocks, pen=make_divisible(1280 * width_mult, divisor=1))


def init_state_dict(seed: int, width_mult: float, num_classes: int = 1000):
    a = arch(width_mult)
    g = SeededInit(seed)
    g.conv("stem.conv.weight", (a["stem"], 3, 3, 3))
    g.bn("stem.bn", a["stem"])
    for i, b in enumerate(a["blocks"]):
        p = f"features.{i}"
        if b["e"] != 1:
            g.conv(f"{p}.conv_exp.conv.weight", (b["dw"], b["cin"], 1, 1))
            g.bn(f"{p}.conv_exp.bn", b["dw"])
        g.conv(f"{p}.conv_dw.conv.weight", (b["dw"], 1, 3, 3))
        g.bn(f"{p}.conv_dw.bn", b["dw"])
        if b["rd"]:
            g.conv(f"{p}.se.fc1.weight", (b["rd"], b["dw"], 1, 1))
            g.vec(f"{p}.se.fc1.bias", b["rd"], "normal", 0.0, 0.1)
            g.bn(f"{p}.se.bn", b["rd"])
            g.conv(f"{p}.se.fc2.weight", (b["dw"], b["rd"], 1, 1))
            g.vec(f"{p}.se.fc2.bias", b["dw"], "normal", 0.0, 0.1)
        g.conv(f"{p}.conv_pwl.conv.weight", (b["cout"], b["dw"], 1, 1), gain=1.0)
        g.bn(f"{p}.conv_pwl.bn", b["cout"])
    n = len(a["blocks"])
    g.conv(f"features.{n}.conv.weight", (a["pen"], a["blocks"][-1]["cout"], 1, 1))
    g.bn(f"features.{n}.bn", a["pen"])
    if num_classes > 0:
        g.conv("head.fc.weight", (num_classes, a["pen"]), gain=1.0)
        g.vec("head.fc.bias", num_classes, "normal", 0.0, 0.1)
    return g.sd


def _cba(x, sd, p, rb, stride=1, pad=0, groups=1):
    w, b = fold_bn(sd[f"{p}.conv.weight"], bn_of(sd, f"{p}.bn"), BN_EPS)
    return F.conv2d(x, rb(w), b, stride=stride, padding=pad, groups=groups)


def forward_features(sd, x, width_mult, sim_bf16=False, taps=None):
    """(B,3,H,W) -> un-pooled (B, pen_chs, H/32, W/32).  ``sim_bf16``: the HIP path's rounding points."""
    rb = Rounder(sim_bf16)
    a = arch(width_mult)
    x = rb(F.silu(_cba(x, sd, "stem", rb, stride=2, pad=1)))
    if taps is not None:
        taps["stem"] = x
    for i, b in enumerate(a["blocks"]):
        p = f"features.{i}"
        sc = x
        if b["e"] != 1:
            x = rb(F.silu(_cba(x, sd, f"{p}.conv_exp", rb)))
        d = _cba(x, sd, f"{p}.conv_dw", rb, stride=b["s"], pad=1, groups=b["dw"])   # BN, no activation
        if b["rd"]:
            s = d.mean((2, 3), keepdim=True)
            bn = bn_of(sd, f"{p}.se.bn")
            if sim_bf16:
                # the HIP path folds the SE BN into fc1 and stores both FC matrices rounded to bf16 (fp32 math)
                scale = bn["weight"] / torch.sqrt(bn["running_var"] + BN_EPS)
                w1 = rb(sd[f"{p}.se.fc1.weight"] * scale.reshape(-1, 1, 1, 1))
                b1 = sd[f"{p}.se.fc1.bias"] * scale + (bn["bias"] - bn["running_mean"] * scale)
                r = F.relu(F.conv2d(s, w1, b1))
                gate = torch.sigmoid(F.conv2d(r, rb(sd[f"{p}.se.fc2.weight"]), sd[f"{p}.se.fc2.bias"]))
            else:
                r = F.conv2d(s, sd[f"{p}.se.fc1.weight"], sd[f"{p}.se.fc1.bias"])
                r = F.relu(F.batch_norm(r, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, BN_EPS))
                gate = torch.sigmoid(F.conv2d(r, sd[f"{p}.se.fc2.weight"], sd[f"{p}.se.fc2.bias"]))
            x = rb(F.relu6(rb(d) * gate))
        else:
            x = rb(F.relu6(rb(d)))
        x = _cba(x, sd, f"{p}.conv_pwl", rb)
        if b["s"] == 1 and b["cin"] <= b["cout"]:
            x = torch.cat([x[:, : b["cin"]] + sc, x[:, b["cin"]:]], 1)   # x[:, 0:in_chs] += shortcut
        x = rb(x)
        if taps is not None:
            taps[p] = x
    n = len(a["blocks"])
    return rb(F.silu(_cba(x, sd, f"features.{n}", rb)))


def forward(sd, x, width_mult, sim_bf16=False):
    rb = Rounder(sim_bf16)
    fm = forward_features(sd, x, width_mult, sim_bf16)
    f = fm.mean((2, 3))
    if "head.fc.weight" in sd:
        return F.linear(rb(f), rb(sd["head.fc.weight"]), sd["head.fc.bias"])
    return f
