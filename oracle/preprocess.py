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
