"""Inference-time pre-processing on the GPU: the transform chain of inference/inference.py:48-52
``Compose([SquarePad(), ToTensor(), Normalize(ImageNet mean/std)])`` (SquarePad: utils/square_pad.py:20-36) for uint8
images that are already on the device, written straight into the model's NCHW fp32 input batch."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import MI355Error, check, lib, require_cuda, stream_ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def square_pad_normalize(images, mean=IMAGENET_MEAN, std=IMAGENET_STD, fill: int = 255) -> torch.Tensor:
    """``images``: a list of uint8 (H, W, 3) device tensors whose longer side is the same S -> (B, 3, S, S) fp32."""
    if not images:
        raise MI355Error("square_pad_normalize needs at least one image")
    S = max(int(max(im.shape[0], im.shape[1])) for im in images)
    dev = images[0].device
    out = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=dev)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    with torch.cuda.device(dev):
        for b, im in enumerate(images):
            require_cuda(im, "image")
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise MI355Error(f"image {b}: expected uint8 (H, W, 3), got {im.dtype} {tuple(im.shape)}")
            if max(im.shape[0], im.shape[1]) != S:
                raise MI355Error(f"image {b}: longer side {max(im.shape[0], im.shape[1])} != {S}; resize first")
            im = im.contiguous()
            check(lib().mi355_square_pad_normalize(im.data_ptr(), im.shape[0], im.shape[1], int(fill), m, s,
                                                   out[b].data_ptr(), stream_ptr(dev)))
    return out
