"""Pre-processing on the GPU for uint8 images that are already on the device, written straight
into the model's NCHW fp32 input batch:
  * inference/inference.py:48-52  ``Compose([SquarePad(), ToTensor(), Normalize(ImageNet mean/std)])``
    (SquarePad: utils/square_pad.py:20-36)                                   -> ``square_pad_normalize``
  * train/train.py:48-50          ``Compose([Resize((224, 224)), ToTensor()])`` -> ``resize`` / ``resize_to_tensor``
    (Resize on a PIL image is Pillow's antialiased BILINEAR resample, reproduced bit-exactly)."""
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


def resize(image: torch.Tensor, size=(224, 224)) -> torch.Tensor:
    """``transforms.Resize(size)(pil_image)`` for a uint8 (H, W, 3) device tensor -> uint8 (size[0], size[1], 3);
    ``size`` is (h, w) as in torchvision.  Bit-exact with ``PIL.Image.resize((w, h), Image.BILINEAR)``."""
    require_cuda(image, "image")
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise MI355Error(f"resize: expected uint8 (H, W, 3), got {image.dtype} {tuple(image.shape)}")
    if isinstance(size, int):
        raise MI355Error("resize: size must be (h, w); the smaller-edge form of torchvision.Resize(int) is not used by the reference")
    oh, ow = int(size[0]), int(size[1])
    im = image.contiguous()
    h, w = int(im.shape[0]), int(im.shape[1])
    out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=im.device)
    tmp = torch.empty((h * ow * 3,), dtype=torch.uint8, device=im.device) if (h != oh and w != ow) else None
    with torch.cuda.device(im.device):
        check(lib().mi355_resize_bilinear_u8(im.data_ptr(), h, w, out.data_ptr(), oh, ow,
                                             tmp.data_ptr() if tmp is not None else None, stream_ptr(im.device)))
    return out


def resize_to_tensor(images, size=(224, 224)) -> torch.Tensor:
    """train/train.py:48: ``Compose([Resize(size), ToTensor()])`` for a list of uint8 (H, W, 3) device images ->
    (B, 3, size[0], size[1]) fp32 in [0, 1] (no normalisation: the training scripts do not normalise)."""
    if int(size[0]) != int(size[1]):
        raise MI355Error("resize_to_tensor: the reference only uses square targets (224, 224)")
    return square_pad_normalize([resize(im, size) for im in images], mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
