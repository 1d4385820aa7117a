"""CPU oracle: timm-0.4.12 ``swin_base_patch4_window7_224`` forward (TEST INFRASTRUCTURE — oracle/__init__.py).

PARITY UNPINNED (timm not vendored/installed; no reference fixture).  Restated from timm 0.4.12's published
``swin_transformer.py``: patch embed 4x4/s4 conv (3->128) + LayerNorm; stages depth (2,2,18,2), heads (4,8,16,32),
window 7, shift 3 on odd blocks (0 where the resolution equals the window), relative position bias table
(13*13, heads) indexed by ``relative_position_index``, -100 attention mask on shifted windows, MLP x4 with exact
GELU, PatchMerging (2x2 concat -> LayerNorm(4C) -> Linear(4C,2C,bias=False)), final LayerNorm, token mean, head.
Anchored by the published parameter count 87 768 224.  Call site: ``timm.create_model(...)`` +
``model.head = Identity()`` at train/train_vit_triplet.py:354-357.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .common import Rounder, SeededInit

EMBED, DEPTHS, HEADS, WS, IMG, PATCH = 128, (2, 2, 18, 2), (4, 8, 16, 32), 7, 224, 4
LN_EPS = 1e-5


def relative_position_index(ws=WS):
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def attn_mask(H, W, ws, shift):
    img_mask = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img_mask, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


def layout():
    """[(stage, block, dim, heads, res, shift)], with timm's rule: window = min(res, 7), shift 0 when res <= 7."""
    out = []
    for s, depth in enumerate(DEPTHS):
        dim, res = EMBED * 2 ** s, IMG // PATCH // 2 ** s
        for b in range(depth):
            shift = 0 if (b % 2 == 0 or res <= WS) else WS // 2
            out.append((s, b, dim, HEADS[s], res, shift))
    return out


def init_state_dict(seed: int, num_classes: int = 1000):
    g = SeededInit(seed)
    g.conv("patch_embed.proj.weight", (EMBED, 3, PATCH, PATCH), gain=1.0)
    g.vec("patch_embed.proj.bias", EMBED, "normal", 0.0, 0.1)
    g.ln("patch_embed.norm", EMBED)
    for (s, b, dim, nh, res, shift) in layout():
        p = f"layers.{s}.blocks.{b}"
        if shift > 0:
            g.sd[f"{p}.attn_mask"] = attn_mask(res, res, WS, shift)
        g.ln(f"{p}.norm1", dim)
        g.vec(f"{p}.attn.relative_position_bias_table", (2 * WS - 1) ** 2 * nh, "normal", 0.0, 0.2)
        g.sd[f"{p}.attn.relative_position_bias_table"] = g.sd[f"{p}.attn.relative_position_bias_table"].view(-1, nh)
        g.sd[f"{p}.attn.relative_position_index"] = relative_position_index()
        g.conv(f"{p}.attn.qkv.weight", (3 * dim, dim), gain=1.0)
        g.vec(f"{p}.attn.qkv.bias", 3 * dim, "normal", 0.0, 0.1)
        g.conv(f"{p}.attn.proj.weight", (dim, dim), gain=0.25)
        g.vec(f"{p}.attn.proj.bias", dim, "normal", 0.0, 0.02)
        g.ln(f"{p}.norm2", dim)
        g.conv(f"{p}.mlp.fc1.weight", (4 * dim, dim), gain=2.0)
        g.vec(f"{p}.mlp.fc1.bias", 4 * dim, "normal", 0.0, 0.1)
        g.conv(f"{p}.mlp.fc2.weight", (dim, 4 * dim), gain=0.25)
        g.vec(f"{p}.mlp.fc2.bias", dim, "normal", 0.0, 0.02)
        if b == DEPTHS[s] - 1 and s < len(DEPTHS) - 1:
            g.conv(f"layers.{s}.downsample.reduction.weight", (2 * dim, 4 * dim), gain=1.0)
            g.ln(f"layers.{s}.downsample.norm", 4 * dim)
    g.ln("norm", EMBED * 8)
    if num_classes > 0:
        g.conv("head.weight", (num_classes, EMBED * 8), gain=1.0)
        g.vec("head.bias", num_classes, "normal", 0.0, 0.1)
    return g.sd


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], LN_EPS)


def forward_features(sd, x, sim_bf16=False, taps=None):
    """(B,3,224,224) -> pooled (B,1024), = timm swin ``forward_features``.  ``sim_bf16`` rounds where the HIP path
    rounds: linear weights, every stored activation (LN outputs, qkv, attention probabilities, attention output,
    residual stream, MLP hidden); LN statistics, softmax, GELU and accumulation stay fp32."""
    rb = Rounder(sim_bf16)
    B = x.shape[0]
    x = F.conv2d(x, rb(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    x = rb(_ln(x, sd, "patch_embed.norm"))
    if taps is not None:
        taps["patch_embed"] = x
    for (s, b, dim, nh, res, shift) in layout():
        p = f"layers.{s}.blocks.{b}"
        H = W = res
        shortcut = x
        h = rb(_ln(x, sd, f"{p}.norm1")).view(B, H, W, dim)
        if shift > 0:
            h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
        win = window_partition(h, WS).view(-1, WS * WS, dim)
        Bw, N, C = win.shape
        qkv = rb(F.linear(win, rb(sd[f"{p}.attn.qkv.weight"]), sd[f"{p}.attn.qkv.bias"]))
        qkv = qkv.reshape(Bw, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q * (C // nh) ** -0.5) @ k.transpose(-2, -1)
        bias = sd[f"{p}.attn.relative_position_bias_table"][sd[f"{p}.attn.relative_position_index"].view(-1)]
        attn = attn + bias.view(N, N, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
        if shift > 0:
            m = sd[f"{p}.attn_mask"]
            nW = m.shape[0]
            attn = attn.view(Bw // nW, nW, nh, N, N) + m.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, nh, N, N)
        attn = rb(torch.softmax(attn, dim=-1))
        o = rb((attn @ v).transpose(1, 2).reshape(Bw, N, C))
        o = window_reverse(o.view(-1, WS, WS, C), WS, H, W)
        if shift > 0:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        o = o.view(B, H * W, C)
        x = rb(shortcut + F.linear(o, rb(sd[f"{p}.attn.proj.weight"]), sd[f"{p}.attn.proj.bias"]))
        hdn = rb(F.gelu(F.linear(rb(_ln(x, sd, f"{p}.norm2")), rb(sd[f"{p}.mlp.fc1.weight"]), sd[f"{p}.mlp.fc1.bias"])))
        x = rb(x + F.linear(hdn, rb(sd[f"{p}.mlp.fc2.weight"]), sd[f"{p}.mlp.fc2.bias"]))
        if taps is not None:
            taps[p] = x
        if b == DEPTHS[s] - 1 and s < len(DEPTHS) - 1:
            xv = x.view(B, H, W, dim)
            xm = torch.cat([xv[:, 0::2, 0::2], xv[:, 1::2, 0::2], xv[:, 0::2, 1::2], xv[:, 1::2, 1::2]], -1)
            xm = rb(_ln(xm.view(B, -1, 4 * dim), sd, f"layers.{s}.downsample.norm"))
            x = rb(F.linear(xm, rb(sd[f"layers.{s}.downsample.reduction.weight"])))
            if taps is not None:
                taps[f"layers.{s}.downsample"] = x
    x = _ln(x, sd, "norm")
    return x.mean(1)


def forward(sd, x, sim_bf16=False):
    rb = Rounder(sim_bf16)
    f = forward_features(sd, x, sim_bf16)
    if "head.weight" in sd:
        return F.linear(rb(f), rb(sd["head.weight"]), sd["head.bias"])
    return f
