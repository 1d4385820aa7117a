"""Shared helpers for the backbone oracles (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from imageretrievalresearch_amd import synth  # data generator only (no compute path)


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """timm 0.4.12 ``make_divisible`` (models/layers/helpers.py): channel rounding."""
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


class Rounder:
    """bf16 rounding points of the HIP path.  ``sim=False`` -> identity (pure fp32 = the
    reference's CPU semantics, autocast is a no-op without CUDA: inference/inference.py:196)."""

    def __init__(self, sim: bool):
        self.sim = sim

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.sim else x


def fold_bn(w: torch.Tensor, bn: dict, eps: float):
    """conv weight (O, ...) + eval-mode BN -> (w * g/sqrt(v+eps), b - m*g/sqrt(v+eps))."""
    scale = bn["weight"] / torch.sqrt(bn["running_var"] + eps)
    shape = [-1] + [1] * (w.dim() - 1)
    return w * scale.reshape(shape), bn["bias"] - bn["running_mean"] * scale


def bn_of(sd, prefix):
    return {k: sd[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


class SeededInit:
    """Seeded random-init weights (SURVEY §8d cfg 1): conv/linear N(0, 2/fan_in), BN gamma
    U[0.5,1.5], beta N(0,0.1), running_mean N(0,0.1), running_var U[0.5,1.5].  Each tensor is
    its own stream ``seed*100003 + ordinal`` of the portable generator, so the same state dict
    can be rebuilt anywhere without torch RNG."""

    def __init__(self, seed: int):
        self.seed = seed
        self.n = 0
        self.sd = OrderedDict()

    def _next(self):
        self.n += 1
        return self.seed * 100003 + self.n

    def conv(self, name, shape, fan_in=None, gain=2.0):
        fan_in = fan_in or int(np.prod(shape[1:]))
        w = synth.normal(self._next(), shape) * np.float32(math.sqrt(gain / fan_in))
        self.sd[name] = torch.from_numpy(w.astype(np.float32))

    def vec(self, name, n, kind, lo=0.0, scale=1.0):
        if kind == "normal":
            v = synth.normal(self._next(), (n,)) * np.float32(scale) + np.float32(lo)
        else:
            v = synth.uniform(self._next(), (n,)) * np.float32(scale) + np.float32(lo)
        self.sd[name] = torch.from_numpy(v.astype(np.float32))

    def bn(self, prefix, n):
        self.vec(f"{prefix}.weight", n, "uniform", 0.5, 1.0)
        self.vec(f"{prefix}.bias", n, "normal", 0.0, 0.1)
        self.vec(f"{prefix}.running_mean", n, "normal", 0.0, 0.1)
        self.vec(f"{prefix}.running_var", n, "uniform", 0.5, 1.0)
        self.sd[f"{prefix}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def ln(self, prefix, n):
        self.vec(f"{prefix}.weight", n, "uniform", 0.5, 1.0)
        self.vec(f"{prefix}.bias", n, "normal", 0.0, 0.1)


def count_params(sd, include_buffers=False):
    tot = 0
    for k, v in sd.items():
        if not include_buffers and (k.endswith("running_mean") or k.endswith("running_var")
                                    or k.endswith("num_batches_tracked")
                                    or k.endswith("relative_position_index") or k.endswith("attn_mask")):
            continue
        tot += v.numel()
    return tot
