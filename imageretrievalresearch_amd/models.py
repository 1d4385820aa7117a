"""timm-shaped model objects backed by libmi355_retrieval.

``create_model(name, pretrained=False, num_classes=1000)`` mirrors ``timm.create_model`` for the four
backbones the reference uses (inference/inference.py:102,110,133,146 ; train/train.py:396 ;
train/train_vit_triplet.py:354).  The returned ``nn.Module`` exposes the timm 0.4.12 surface the
reference touches:

* ``model(x)`` / ``model.forward_features(x)``; ``.eval() .train() .to() .parameters()``
* timm state-dict keys (``load_state_dict(strict=True|False)``), so Lightning checkpoints with the
  ``model.`` prefix stripped load unchanged (inference/inference.py:117-124)
* assignable heads: ``model.classifier`` (efficientnet), ``model.head`` / ``model.head.fc`` (rexnet),
  ``model.head`` (swin); ``model.head = Identity()`` (train_vit_triplet.py:357) and
  ``model.classifier = Linear(...)`` (inference.py:141) are honoured at the next forward.

The parameter table (names, shapes, order) comes from the C library, which owns the architecture
definitions; Python only mirrors it as ``nn.Parameter``s.  The forward pass is inference-only
(the reference's hot path runs under ``torch.no_grad``): it launches the HIP kernels on torch's
current stream and returns tensors without autograd history.  There is no torch fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import operator
import weakref

import torch
import torch.nn as nn

from ._lib import MI355Error, check, lib, require_cuda, stream_ptr

_FAMILY = {
    "efficientnet_b3a": "efficientnet", "efficientnet_b3": "efficientnet",
    "rexnet_100": "rexnet", "rexnet_130": "rexnet", "rexnet_150": "rexnet", "rexnet_200": "rexnet",
    "swin_base_patch4_window7_224": "swin",
}
_HEAD_PATH = {"efficientnet": "classifier", "rexnet": "head.fc", "swin": "head"}


_VERSION_OF = operator.attrgetter("_version")


def list_models():
    return sorted(_FAMILY)


class _Node(nn.Module):
    """Structural container so dotted timm keys map onto real submodules.  Replacing a parameter / buffer OBJECT on it
    (``model.conv_stem.weight = nn.Parameter(...)``, ``torch.nn.utils.parametrize`` / ``prune``) tells the owning model, whose
    cached tensor list would otherwise keep watching the old tensor while the packed HIP weights go stale."""

    def __getitem__(self, i):          # timm's `blocks` / `features` / `layers` are nn.Sequential: model.blocks[3][2]
        key = str(i if i >= 0 else len(self._modules) + i)
        if key not in self._modules:
            raise IndexError(i)
        return self._modules[key]

    def __len__(self):
        return len(self._modules)

    def _touch_root(self):
        ref = self.__dict__.get("_root")
        root = ref() if ref is not None else None
        if root is not None:
            root.__dict__["_sig_tensors"] = None
            root.__dict__["_dirty"] = True

    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, nn.Module)) or name in self._parameters or name in self._buffers:
            self._touch_root()
        super().__setattr__(name, value)

    def __delattr__(self, name):
        self._touch_root()
        super().__delattr__(name)

    def register_parameter(self, name, param):
        self._touch_root()
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        self._touch_root()
        super().register_buffer(name, tensor, persistent=persistent)


class ClassifierHead(nn.Module):
    """timm's rexnet ``ClassifierHead``: global-avg-pool + fc; callable on the un-pooled map
    (train/train.py:194-195 ``fm = self.model.forward_features(x); lbl = self.model.head(fm)``).  The call runs in the HIP
    library (``mi355_pool_linear``: pool + Linear in one kernel, same rounding points as the classifier inside
    ``model(x)``); like every op of the package it has no CPU path."""

    def __init__(self, in_features, num_classes):
        super().__init__()
        self.fc = nn.Linear(in_features, num_classes) if num_classes > 0 else nn.Identity()

    def forward(self, x):
        # The HIP head has no backward.  A training step (train/train.py:194-195 computes a classification loss on these
        # logits) must not lose the head's gradient silently: refuse it the way the model's own forward does.
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.fc.parameters()))):
            raise MI355Error("ClassifierHead runs in the HIP library and is forward-only (no autograd history): call it under "
                             "torch.no_grad() / in eval mode, or train the head with torch ops on get_fm(fm)")
        return pool_linear(x, self.fc)


def pool_linear(fm: torch.Tensor, fc: "nn.Module | None" = None) -> torch.Tensor:
    """``get_fm`` (train/train.py:84-103) when ``fc`` is None / Identity, else ``fc(get_fm(fm))`` for an nn.Linear ``fc``:
    fm (B, C, H, W) fp32 on the GPU -> (B, C) or (B, N)."""
    require_cuda(fm, "feature map")
    if fm.dim() != 4:
        raise MI355Error(f"expected an un-pooled (B, C, H, W) map, got {tuple(fm.shape)}")
    fm = fm.detach().float().contiguous()
    B, Cc, H, W = fm.shape
    lin = fc if isinstance(fc, nn.Linear) else None
    if fc is not None and lin is None and not isinstance(fc, nn.Identity):
        raise MI355Error("pool_linear: the head must be nn.Linear or nn.Identity")
    if lin is not None and lin.in_features != Cc:
        raise MI355Error(f"head expects {lin.in_features} features, the map has {Cc}")
    out = torch.empty((B, lin.out_features if lin is not None else Cc), dtype=torch.float32, device=fm.device)
    if B:
        w = lin.weight.detach().to(fm.device, torch.float32).contiguous() if lin is not None else None
        bia = lin.bias.detach().to(fm.device, torch.float32).contiguous() if (lin is not None and lin.bias is not None) else None
        with torch.cuda.device(fm.device):
            check(lib().mi355_pool_linear(fm.data_ptr(), B, Cc, H * W, w.data_ptr() if w is not None else None,
                                          bia.data_ptr() if bia is not None else None,
                                          lin.out_features if lin is not None else 0,
                                          out.data_ptr() if lin is not None else None,
                                          out.data_ptr() if lin is None else None, stream_ptr(fm.device)))
    return out


def _swin_relative_position_index(ws: int) -> torch.Tensor:
    """timm WindowAttention's ``relative_position_index`` buffer (kept only for state-dict fidelity: the HIP path
    bakes the dense bias at pack time)."""
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _swin_attn_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    """timm SwinTransformerBlock's ``attn_mask`` buffer (0 / -100); the kernel derives it from region labels."""
    img = torch.zeros((res, res))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[h, w] = cnt
            cnt += 1
    mw = img.view(res // ws, ws, res // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def _table(handle):
    L = lib()
    out = []
    name = C.c_char_p()
    ndim = C.c_int()
    shape = (C.c_int64 * 4)()
    kind = C.c_int()
    for i in range(L.mi355_model_num_tensors(handle)):
        check(L.mi355_model_tensor_info(handle, i, C.byref(name), C.byref(ndim), shape, C.byref(kind)))
        out.append((name.value.decode(), tuple(shape[d] for d in range(ndim.value)), kind.value))
    return out


def _new_handle(name: str, num_classes: int):
    h = C.c_void_p()
    check(lib().mi355_model_create(name.encode(), int(num_classes), C.byref(h)))
    return h


class MI355Model(nn.Module):
    def __init__(self, model_name: str, num_classes: int = 1000, seed: int = 0):
        super().__init__()
        if model_name not in _FAMILY:
            # same guard wording as the reference (train/train.py:400)
            raise AssertionError(f"Unknown model name {model_name!r}; known: {list_models()}")
        self.model_name = model_name
        self.family = _FAMILY[model_name]
        self.__dict__["_handle"] = None
        self.__dict__["_handle_classes"] = None
        self.__dict__["_dirty"] = True
        self.__dict__["_sig"] = None
        self.__dict__["_pack_device"] = None
        handle = _new_handle(model_name, num_classes)
        self.__dict__["_handle"] = handle
        self.__dict__["_handle_classes"] = num_classes
        self.num_features = lib().mi355_model_feature_dim(handle)
        self.num_classes = num_classes
        head_prefix = _HEAD_PATH[self.family] + "."
        for name, shape, kind in _table(handle):
            if name.startswith(head_prefix):
                continue  # the classifier is a real nn.Linear, created below
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    node = _Node()
                    node.__dict__["_root"] = weakref.ref(self)
                    mod.add_module(p, node)
                mod = mod._modules[p]
            if kind == 0:
                mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
            elif kind == 1:
                mod.register_buffer(parts[-1], torch.zeros(shape))
            else:
                mod.register_buffer(parts[-1], torch.zeros(shape, dtype=torch.int64))
        D = self.num_features
        if self.family == "efficientnet":
            self.classifier = nn.Linear(D, num_classes) if num_classes > 0 else nn.Identity()
        elif self.family == "rexnet":
            self.head = ClassifierHead(D, num_classes)
        else:
            self.head = nn.Linear(D, num_classes) if num_classes > 0 else nn.Identity()
        self.reset_parameters(seed)
        # A parent's load_state_dict (nn.Sequential(conv_input, model) at inference/inference.py:103-124, a
        # LightningModule holding self.model) never calls OUR load_state_dict override: it walks the tree and calls every
        # module's _load_from_state_dict, which runs the module's pre-hooks - so the re-pack is requested from one.
        self._register_load_state_dict_pre_hook(self._on_load_state_dict)

    def _on_load_state_dict(self, *args, **kwargs):
        self.__dict__["_dirty"] = True
        self.__dict__["_sig_tensors"] = None

    # ------------------------------------------------------------------ init
    @torch.no_grad()
    def reset_parameters(self, seed: int = 0):
        """Seeded random init (there are no downloadable weights offline): conv/linear N(0, 2/fan_in),
        norm scales 1, shifts 0, running stats (0, 1)."""
        g = torch.Generator().manual_seed(seed)
        for name, t in list(self.named_parameters()) + list(self.named_buffers()):
            if t.dtype != torch.float32:
                if name.endswith("relative_position_index"):
                    t.copy_(_swin_relative_position_index(int(round(t.shape[0] ** 0.5))))
                else:
                    t.zero_()
                continue
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "attn_mask":
                ws = int(round(t.shape[1] ** 0.5))
                res = int(round(t.shape[0] ** 0.5)) * ws
                t.copy_(_swin_attn_mask(res, ws, ws // 2))
            elif leaf == "running_var":
                t.fill_(1.0)
            elif leaf in ("running_mean", "bias"):
                t.zero_()
            elif leaf == "relative_position_bias_table":
                t.copy_(torch.randn(t.shape, generator=g) * 0.02)
            elif t.dim() >= 2:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in))
            else:  # norm weight
                t.fill_(1.0)
        self._dirty = True

    # ------------------------------------------------------------------ dirty tracking
    def _apply(self, fn, *a, **k):
        self.__dict__["_dirty"] = True
        self.__dict__["_sig_tensors"] = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self.__dict__["_dirty"] = True
        self.__dict__["_sig_tensors"] = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def __setattr__(self, name, value):
        if name in ("classifier", "head"):
            self.__dict__["_dirty"] = True
            self.__dict__["_sig_tensors"] = None
        super().__setattr__(name, value)

    def mark_dirty(self):
        """Call after modifying parameters in place in eval mode (train mode re-checks every call)."""
        self._dirty = True

    def _head_linear(self):
        mod = self
        for p in _HEAD_PATH[self.family].split("."):
            mod = getattr(mod, p, None)
            if mod is None:
                return None
        return mod if isinstance(mod, nn.Linear) else None

    def _signature(self):
        """(storage, version) of every parameter and buffer: changes when one is written in place (optimizer step,
        ``param.copy_``) or re-allocated.  The tensor list is cached (walking the module tree is 1.5 ms, this is 0.1 ms);
        everything that REPLACES tensor objects (``_apply``, ``load_state_dict``, head assignment, attribute assignment on
        a sub-module: ``_Node.__setattr__``) drops the cache."""
        ts = self.__dict__.get("_sig_tensors")
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__["_sig_tensors"] = ts
        return tuple(map(torch.Tensor.data_ptr, ts)) + tuple(map(_VERSION_OF, ts))

    def _pack(self, device):
        L = lib()
        fc = self._head_linear()
        ncls = fc.out_features if fc is not None else 0
        if fc is not None and fc.in_features != self.num_features:
            raise MI355Error(f"classifier expects {fc.in_features} features, backbone gives {self.num_features}")
        if self._handle_classes != ncls:
            L.mi355_model_destroy(self._handle)
            self.__dict__["_handle"] = _new_handle(self.model_name, ncls)
            self.__dict__["_handle_classes"] = ncls
        self.num_classes = ncls
        sd = self.state_dict()
        head_prefix = _HEAD_PATH[self.family] + "."
        for name, shape, kind in _table(self._handle):
            if kind == 2:
                continue
            if name not in sd:
                raise MI355Error(f"state dict has no tensor {name!r}")
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape) and not (name.startswith(head_prefix)):
                raise MI355Error(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
            check(L.mi355_model_set_tensor(self._handle, name.encode(), t.data_ptr(), t.numel()))
        with torch.cuda.device(device):
            check(L.mi355_model_pack(self._handle, stream_ptr(device)))
        self.__dict__["_dirty"] = False
        self.__dict__["_pack_device"] = device
        self.__dict__["_sig_tensors"] = None
        self.__dict__["_sig"] = self._signature()

    def _ensure_packed(self, device):
        # eval mode too: a reload through a parent module or an in-place write must never leave stale packed weights
        if self._dirty or self._pack_device != device or self._sig != self._signature():
            self._pack(device)

    # ------------------------------------------------------------------ options / introspection
    def set_option(self, key: str, value: int):
        check(lib().mi355_model_set_option(self._handle, key.encode(), int(value)))
        return self

    def traffic(self, B: int, H: int = 224, W: int = 224):
        """Algorithmic bytes / MACs of one forward (layer-granular model)."""
        a, w, m = C.c_double(), C.c_double(), C.c_double()
        check(lib().mi355_model_traffic(self._handle, B, H, W, C.byref(a), C.byref(w), C.byref(m)))
        by = (C.c_double * 8)()
        mc = (C.c_double * 8)()
        check(lib().mi355_model_traffic_kinds(self._handle, B, H, W, by, mc, 8))
        kinds = ["stem", "gemm", "dw", "se", "other", "attn", "ln", "fused"]
        return {"act_bytes": a.value, "weight_bytes": w.value, "macs": m.value,
                "bytes_by_kind": dict(zip(kinds, list(by))), "macs_by_kind": dict(zip(kinds, list(mc)))}

    def profile_read(self):
        ms = (C.c_double * 8)()
        n = (C.c_int64 * 8)()
        check(lib().mi355_model_profile_read(self._handle, ms, n, 8))
        kinds = ["stem", "gemm", "dw", "se", "other", "attn", "ln", "fused"]
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(kinds)}

    def profile_ops(self, B: int, H: int = 224, W: int = 224):
        """Per-op table [(label, kind, avg_ms, algorithmic_bytes)] after ``profile_read()``."""
        n = 1024
        ms = (C.c_double * n)()
        by = (C.c_double * n)()
        kd = (C.c_int * n)()
        lab = C.create_string_buffer(n * 64)
        cnt = lib().mi355_model_profile_ops(self._handle, B, H, W, n, ms, by, kd, lab, 64)
        kinds = ["stem", "gemm", "dw", "se", "other", "attn", "ln", "fused"]
        return [(lab.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode(), kinds[kd[i]], ms[i], by[i]) for i in range(cnt)]

    def block_stamps(self):
        """Per-op phase cycle counts of the whole-block kernel (after ``set_option('block_stamps', 1)`` and a forward):
        list of (op_index, [16 cycle buckets]) for the ops that ran as a block (bucket list: end of k_mbconv_block)."""
        n = 1024
        out = (C.c_double * (n * 16))()
        cnt = lib().mi355_model_block_stamps(self._handle, out, n)
        if cnt < 0:
            check(1)
        return [(i, [out[i * 16 + j] for j in range(16)]) for i in range(cnt) if sum(out[i * 16:i * 16 + 16]) > 0]

    def run_between_taps(self, from_tap: str, to_tap: str, x: torch.Tensor):
        """Parity tool: run only the layers behind ``from_tap`` up to ``to_tap`` on ``x`` (B,C,h,w fp32 on the GPU: the
        oracle's tap of the previous layer); read the results with ``read_tap`` (taps must be enabled)."""
        require_cuda(x, "activation")
        x = x.detach().float().contiguous()
        self._ensure_packed(x.device)
        B, Cc, h, w = x.shape
        with torch.cuda.device(x.device):
            check(lib().mi355_model_run_between_taps(self._handle, from_tap.encode(), to_tap.encode(), x.data_ptr(),
                                                     B, Cc, h, w, stream_ptr(x.device)))
        return self

    def enable_taps(self, on: bool = True):
        check(lib().mi355_model_enable_taps(self._handle, int(on)))
        return self

    def read_tap(self, name: str) -> torch.Tensor:
        shape = (C.c_int64 * 4)()
        check(lib().mi355_model_read_tap(self._handle, name.encode(), None, 0, shape, None))
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self._pack_device)
        with torch.cuda.device(out.device):
            check(lib().mi355_model_read_tap(self._handle, name.encode(), out.data_ptr(), out.numel(), shape,
                                             stream_ptr(out.device)))
        return out

    # ------------------------------------------------------------------ forward
    def _refuse_training(self):
        """The reference's training loops call the same Module (train/train.py:136,194): they would get eval-mode BatchNorm
        outputs with no autograd history and train nothing.  Out of scope to implement, in scope to refuse."""
        if self.training and torch.is_grad_enabled():
            raise MI355Error(f"{self.model_name}: this backbone is forward-only (HIP kernels, eval-mode BatchNorm, no autograd "
                             "history); call model.eval() and/or run it under torch.no_grad() - training steps such as "
                             "train/train.py:194 are not supported")

    def _custom_head(self, out):
        """A head that is neither Linear nor Identity (user-assigned module) runs on the pooled features."""
        head = self
        for p in _HEAD_PATH[self.family].split("."):
            nxt = getattr(head, p, None)
            if nxt is None:        # e.g. rexnet `model.head = Identity()`: there is no `.fc` below it
                break
            head = nxt
        if not isinstance(head, (nn.Linear, nn.Identity, ClassifierHead)):
            out = head(out)
        return out

    def _prep(self, x):
        self._refuse_training()
        require_cuda(x, "model input")
        if x.dim() != 4 or x.shape[1] != 3:
            raise MI355Error(f"expected (B,3,H,W) input, got {tuple(x.shape)}")
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        self._ensure_packed(x.device)
        return x

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        """timm ``forward_features``: un-pooled (B,C,H/32,W/32) for efficientnet/rexnet, pooled (B,C) for swin."""
        x = self._prep(x)
        B, _, H, W = x.shape
        D = self.num_features
        if self.family == "swin":
            out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((B, D, (H + 31) // 32, (W + 31) // 32), dtype=torch.float32, device=x.device)
        if B:
            with torch.cuda.device(x.device):
                check(lib().mi355_model_forward_features(self._handle, x.data_ptr(), B, H, W, out.data_ptr(), None,
                                                         stream_ptr(x.device)))
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self._prep(x)
        B, _, H, W = x.shape
        n = self.num_classes if self.num_classes > 0 else self.num_features
        out = torch.empty((B, n), dtype=torch.float32, device=x.device)
        if B:
            with torch.cuda.device(x.device):
                check(lib().mi355_model_forward(self._handle, x.data_ptr(), B, H, W, out.data_ptr(), None,
                                                stream_ptr(x.device)))
        return self._custom_head(out)

    def forward_uint8(self, images: torch.Tensor, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                      fill: int = 255, conv_input: "nn.Module | None" = None, features: bool = False) -> torch.Tensor:
        """The reference's whole inference front end in ONE kernel in front of the backbone: ``images`` (B, h, w, 3) uint8
        on the GPU (one size per batch) -> SquarePad(fill) -> ToTensor -> Normalize(mean, std) (inference/inference.py:48-52)
        -> ``conv_input`` (the ``Sequential(Conv2d(3,3,3,1,1,bias=False), SiLU)`` of inference/inference.py:103-105, or
        None) -> stem ... -> ``forward`` (or ``forward_features`` with ``features=True``).  No fp32 NCHW batch is ever
        written; bit-identical to ``preprocess.square_pad_normalize`` + ``conv_input`` + ``forward``.  Swin fuses the
        transform into its 4x4 patch embedding the same way (images whose longer side is 224, no conv_input)."""
        self._refuse_training()
        require_cuda(images, "images")
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise MI355Error(f"forward_uint8 expects uint8 (B, h, w, 3), got {images.dtype} {tuple(images.shape)}")
        if self.family == "swin" and conv_input is not None:
            raise MI355Error("forward_uint8: conv_input belongs to the convolutional backbones (swin fuses the transform into its patch embedding)")
        images = images.contiguous()
        self._ensure_packed(images.device)
        B, h, w, _ = images.shape
        S = max(h, w)
        cw = None
        if conv_input is not None:
            conv = conv_input[0] if isinstance(conv_input, nn.Sequential) else getattr(conv_input, "conv", conv_input)
            if not isinstance(conv, nn.Conv2d) or tuple(conv.weight.shape) != (3, 3, 3, 3) or conv.bias is not None:
                raise MI355Error("conv_input must be Sequential(Conv2d(3, 3, 3, 1, 1, bias=False), SiLU)")
            cw = conv.weight.detach().to(images.device, torch.float32).contiguous()
        D = self.num_features
        if features and self.family == "swin":
            out = torch.empty((B, D), dtype=torch.float32, device=images.device)      # timm's swin forward_features is pooled
        elif features:
            out = torch.empty((B, D, (S + 31) // 32, (S + 31) // 32), dtype=torch.float32, device=images.device)
        else:
            out = torch.empty((B, self.num_classes if self.num_classes > 0 else D), dtype=torch.float32, device=images.device)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        if B:
            with torch.cuda.device(images.device):
                check(lib().mi355_model_forward_u8(self._handle, images.data_ptr(), B, h, w, int(fill), m3, s3,
                                                   cw.data_ptr() if cw is not None else None, int(features),
                                                   out.data_ptr(), None, stream_ptr(images.device)))
        return out if features else self._custom_head(out)

    def embed(self, x: torch.Tensor):
        """Pooled embeddings (B, D) regardless of the head — ``get_fm(forward_features(x))`` of
        train/train.py:84-103 in one call — plus the logits when a classifier is attached."""
        x = self._prep(x)
        B, _, H, W = x.shape
        D = self.num_features
        pooled = torch.empty((B, D), dtype=torch.float32, device=x.device)
        n = self.num_classes if self.num_classes > 0 else D
        out = torch.empty((B, n), dtype=torch.float32, device=x.device)
        if B:
            with torch.cuda.device(x.device):
                check(lib().mi355_model_forward(self._handle, x.data_ptr(), B, H, W, out.data_ptr(),
                                                pooled.data_ptr(), stream_ptr(x.device)))
        return pooled, (out if self.num_classes > 0 else None)

    def __del__(self):
        h = self.__dict__.get("_handle")
        if h is not None:
            try:
                lib().mi355_model_destroy(h)
            except Exception:
                pass


def create_model(model_name: str, pretrained: bool = False, num_classes: int = 1000, seed: int = 0, **kwargs):
    """``timm.create_model`` for the reference's backbones.  ``pretrained=True`` would fetch ImageNet
    weights from the network in timm (train/train.py:396); there is no network here, so it raises."""
    if pretrained:
        raise MI355Error("pretrained=True needs a download, which is unavailable offline; create the model with "
                         "pretrained=False and load a checkpoint with load_state_dict")
    if kwargs:
        raise TypeError(f"unsupported create_model arguments: {sorted(kwargs)}")
    return MI355Model(model_name, num_classes=num_classes, seed=seed)


class ConvInput(nn.Sequential):
    """The reference's pre-stem ``Sequential(Conv2d(3,3,3,1,1,bias=False), SiLU)``
    (inference/inference.py:103-104) with the same state-dict keys, run as one HIP kernel."""

    def __init__(self):
        super().__init__(nn.Conv2d(3, 3, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
                         nn.SiLU(inplace=True))

    def forward(self, x):
        require_cuda(x, "conv_input input")
        x = x.detach().float().contiguous()
        w = self[0].weight.detach().to(x.device, torch.float32).contiguous()   # weight may still live on the CPU
        out = torch.empty_like(x)
        B, _, H, W = x.shape
        if B:
            with torch.cuda.device(x.device):
                check(lib().mi355_conv_input_silu(x.data_ptr(), w.data_ptr(), B, H, W, out.data_ptr(),
                                                  stream_ptr(x.device)))
        return out


def with_conv_input(base_model: nn.Module) -> nn.Sequential:
    """``Sequential(conv_layer, base_model)`` of inference/inference.py:105 (keys ``0.0.weight``, ``1.*``)."""
    return nn.Sequential(ConvInput(), base_model)
