"""Portable counter-based synthetic data generator (SURVEY.md §8d "portable synthetic generator").

Every element is a pure function of ``(seed, flat_index)`` through splitmix64, using only
integer arithmetic plus ONE float32 multiply, so numpy (fixtures, CPU oracle) and the HIP
kernel ``mi355_synth_fill`` (csrc/synth.hip) produce bit-identical tensors on any machine and
any torch/numpy version.  No libm calls (log/cos of Box-Muller are not bit-reproducible
between host libm and the GPU), so the "normal" stream is an Irwin-Hall(4) sum of the four
16-bit fields of the hash, centred and scaled to unit variance.

kind 0: uniform  u = (h >> 40) * 2^-24                       in [0, 1)
kind 1: normal   z = (f0 + f1 + f2 + f3 - 131070) * SCALE     mean 0, var 1, |z| <= 3.4641
"""
from __future__ import annotations

import numpy as np

UNIFORM = 0
NORMAL = 1

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
# var of one 16-bit uniform integer field = (65536^2 - 1) / 12; four of them summed.
NORMAL_SCALE = np.float32(1.0 / np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))
UNIFORM_SCALE = np.float32(2.0 ** -24)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _GOLDEN
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def stream_key(seed: int) -> np.uint64:
    """Per-tensor key: splitmix64(seed), so nearby seeds give unrelated streams."""
    return _splitmix64(np.array([seed], dtype=np.uint64))[0]


def fill(seed: int, n: int, kind: int, offset: int = 0, chunk: int = 1 << 22) -> np.ndarray:
    """float32[n] = elements [offset, offset+n) of stream ``seed``."""
    out = np.empty(n, dtype=np.float32)
    key = stream_key(seed)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(offset + s, offset + e, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = _splitmix64(key + idx)
        if kind == UNIFORM:
            out[s:e] = (h >> np.uint64(40)).astype(np.float32) * UNIFORM_SCALE
        elif kind == NORMAL:
            m = np.uint64(0xFFFF)
            t = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48)))
            out[s:e] = (t.astype(np.int64) - 131070).astype(np.float32) * NORMAL_SCALE
        else:
            raise ValueError(f"unknown kind {kind}")
    return out


def uniform(seed: int, shape, offset: int = 0) -> np.ndarray:
    return fill(seed, int(np.prod(shape)), UNIFORM, offset).reshape(shape)


def normal(seed: int, shape, offset: int = 0) -> np.ndarray:
    return fill(seed, int(np.prod(shape)), NORMAL, offset).reshape(shape)
