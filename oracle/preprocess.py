"""CPU oracle for the inference-time pre-processing (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates, in numpy, the transform chain of inference/inference.py:48-52:
    SquarePad()                      utils/square_pad.py:20-36  (FF.pad(image, (hp, vp, hp+hp_rem, vp+vp_rem), 255, 'constant'))
    transforms.ToTensor()            uint8 HWC -> float32 CHW, x / 255
    transforms.Normalize(mean, std)  (x - mean[c]) / std[c]
torchvision is not installed here, so this is pinned only by construction (PARITY UNPINNED against torchvision itself);
every step is exact integer / single-rounding fp32 arithmetic, which the test restates independently with PIL for the pad.
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def square_pad(img: np.ndarray, fill: int = 255) -> np.ndarray:
    """img uint8 (H, W, 3) -> (S, S, 3), S = max(H, W); utils/square_pad.py:22-36."""
    h, w = img.shape[:2]
    s = max(h, w)
    hp, hp_rem = int((s - w) / 2), (s - w) % 2
    vp, vp_rem = int((s - h) / 2), (s - h) % 2
    out = np.full((s, s, img.shape[2]), fill, dtype=np.uint8)
    out[vp:vp + h, hp:hp + w] = img
    assert out.shape[0] == vp + h + vp + vp_rem and out.shape[1] == hp + w + hp + hp_rem
    return out


def to_tensor_normalize(img: np.ndarray, mean=MEAN, std=STD) -> np.ndarray:
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)          # ToTensor
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)                                     # Normalize: sub_ then div_


def preprocess(img: np.ndarray, mean=MEAN, std=STD) -> np.ndarray:
    return to_tensor_normalize(square_pad(img), mean, std)


# ------------------------------------------------------------------------------------------------
# transforms.Resize((224, 224)) of the training scripts (train/train.py:48-50).  On a PIL image torchvision calls
# ``img.resize((w, h), Image.BILINEAR)``, i.e. Pillow's two-pass antialiased resample (third-party dependency, not in
# /root/reference: Pillow, src/libImaging/Resample.c; Pillow 12.2.0 is installed here and PINS this restatement in
# tests/test_preprocess.py).  Published algorithm, 8-bit path:
#   per axis: scale = in/out, filterscale = max(scale, 1), support = 1.0 * filterscale (triangle filter),
#   for output x: center = (x + 0.5) * scale, xmin = int(center - support + 0.5) clamped at 0,
#                 xmax = int(center + support + 0.5) clamped at in; w_i = tri((i + xmin - center + 0.5) / filterscale),
#                 normalised to sum 1 in double, then fixed point: int(+-0.5 + w * 2^22);
#   a pixel = clip8((2^21 + sum(src_i * k_i)) >> 22); horizontal pass first (only the rows the vertical pass needs),
#   its uint8 result feeds the vertical pass.
# ------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out][2] = (first source index, count), kk int32 [out][ksize])"""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # box is (0, in_size) as C floats
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast of a positive-or-small-negative double: truncation
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """Resample along axis 0 (rows) of a (N, ...) uint8 array with the fixed-point coefficients."""
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for i in range(n):
            acc = acc + src[lo + i] * int(kk[o, i])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 (H, W, C) -> (out_h, out_w, C), bit-exact with PIL.Image.resize((out_w, out_h), Image.BILINEAR)."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    cur = img
    bv, kv = pil_bilinear_coeffs(h, out_h)
    if w != out_w:
        bh, kh = pil_bilinear_coeffs(w, out_w)
        first = int(bv[0, 0]) if h != out_h else 0
        last = int(bv[-1, 0] + bv[-1, 1]) if h != out_h else h
        rows = cur[first:last]                                   # only the rows the vertical pass will read
        cur = _resample_axis0(rows.transpose(1, 0, 2), bh, kh).transpose(1, 0, 2)
        bv = bv.copy()
        bv[:, 0] -= first
    if h != out_h:
        cur = _resample_axis0(cur, bv, kv)
    return cur


def resize_to_tensor(img: np.ndarray, out_h: int = 224, out_w: int = 224) -> np.ndarray:
    """train/train.py:48: Compose([Resize((224, 224)), ToTensor()]) -> float32 (C, out_h, out_w) in [0, 1]."""
    r = pil_resize_bilinear(img, out_h, out_w)
    return (r.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)).astype(np.float32)
