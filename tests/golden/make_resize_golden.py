"""Generate tests/golden/resize_golden.npz from Pillow itself — the third-party library that torchvision's
transforms.Resize((224, 224)) (train/train.py:48-50) calls for a PIL image: PIL.Image.resize((w, h), Image.BILINEAR).
The fixture pins oracle/preprocess.py::pil_resize_bilinear where Pillow is not installed.

usage: python tests/golden/make_resize_golden.py        (Pillow version is recorded in the file)
"""
import os

import numpy as np
import PIL
from PIL import Image

rng = np.random.RandomState(20240807)
yy, xx = np.mgrid[0:157, 0:231]
img = np.stack([(127 + 120 * np.sin(xx / 9.0) * np.cos(yy / 5.0)), (xx * 255.0 / 230), rng.randint(0, 256, size=(157, 231))],
               axis=2).clip(0, 255).astype(np.uint8)
pil = Image.fromarray(img)
out = {
    "img": img,
    "out_224": np.asarray(pil.resize((224, 224), Image.BILINEAR)),     # upscale both sides
    "out_40x64": np.asarray(pil.resize((64, 40), Image.BILINEAR)),     # downscale (antialiased support > 1)
    "pillow_version": np.array(PIL.__version__),
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
