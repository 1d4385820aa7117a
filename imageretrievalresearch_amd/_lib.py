"""ctypes binding of libmi355_retrieval.so (include/mi355_retrieval.h).

The HIP library IS the product path: if it cannot be loaded this module raises — there is no
CPU or eager-torch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_LIB_PATH: another build of the same library (developer A/B runs of two kernel variants on one box)
LIB_PATH = os.environ.get("MI355_LIB_PATH") or os.path.join(_HERE, "libmi355_retrieval.so")

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)
vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/mi355_retrieval.h declares
# (tests/test_abi.py cross-checks this table against the header).
PROTOTYPES = {
    "mi355_abi_version": (C.c_int, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_device_count": (C.c_int, []),
    "mi355_synth_fill": (C.c_int, [vp, C.c_int64, C.c_uint64, C.c_int64, C.c_int, vp]),
    "mi355_l2_normalize_rows": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_float, vp]),
    "mi355_rank_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "mi355_rank_topk": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_int64, vp, vp, vp, C.c_size_t, vp]),
    "mi355_gallery_planes_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "mi355_gallery_prepare": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_size_t, vp]),
    "mi355_rank_topk_prepared": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int64, vp, vp, vp,
                                           C.c_size_t, vp]),
    "mi355_cosine_scores": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_float, vp, vp,
                                      C.c_size_t, vp]),
    "mi355_topk_rows": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int64, vp, vp, vp, C.c_size_t, vp]),
    "mi355_merge_topk": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "mi355_pack_candidates": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    "mi355_merge_packed_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "mi355_merge_packed_topk": (C.c_int, [vp, vp, C.c_int, C.c_int64, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "mi355_pair_cosine": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_float, vp, vp]),
    "mi355_contrastive_loss": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp, vp]),
    "mi355_cosine_embedding_loss": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp]),
    "mi355_hit_counts": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int64, vp, vp]),
    "mi355_distinct_class_topn": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, vp, vp, vp, vp]),
    "mi355_model_create": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "mi355_model_destroy": (None, [vp]),
    "mi355_model_num_tensors": (C.c_int, [vp]),
    "mi355_model_tensor_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "mi355_model_feature_dim": (C.c_int, [vp]),
    "mi355_model_num_classes": (C.c_int, [vp]),
    "mi355_model_set_tensor": (C.c_int, [vp, C.c_char_p, vp, C.c_int64]),
    "mi355_model_pack": (C.c_int, [vp, vp]),
    "mi355_model_forward_features": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "mi355_model_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "mi355_model_forward_u8": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         vp, C.c_int, vp, vp, vp]),
    "mi355_model_enable_taps": (C.c_int, [vp, C.c_int]),
    "mi355_model_read_tap": (C.c_int, [vp, C.c_char_p, vp, C.c_int64, C.POINTER(C.c_int64), vp]),
    "mi355_model_run_between_taps": (C.c_int, [vp, C.c_char_p, C.c_char_p, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi355_model_traffic": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mi355_model_traffic_kinds": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.c_int]),
    "mi355_model_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "mi355_model_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "mi355_model_profile_ops": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "mi355_model_block_stamps": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int]),
    "mi355_pool_linear": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]),
    "mi355_gemm_bf16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi355_square_pad_normalize": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp]),
    "mi355_conv_input_silu": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "mi355_resize_bilinear_u8": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "mi355_score_boost": (C.c_int, [vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp]),
}

_lib = None


class MI355Error(RuntimeError):
    """A libmi355_retrieval call returned nonzero; the message is mi355_last_error()."""


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C imageretrievalresearch_amd/csrc`). There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        if L.mi355_abi_version() != 3:
            raise ImportError(f"ABI version mismatch: library {L.mi355_abi_version()} != binding 3")
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != 0:
        raise MI355Error(lib().mi355_last_error().decode("utf-8", "replace"))


def stream_ptr(device=None) -> int:
    """Raw hipStream_t of torch's CURRENT stream, so results order correctly behind `.item()`."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, name: str):
    if not t.is_cuda:
        raise MI355Error(f"{name} must live on the GPU (got device {t.device}); "
                         "the MI355X path has no CPU fallback")
