"""imageretrievalresearch_amd — MI355X-native embed-then-rank hot path.

A drop-in for ONE path of vitasoftAI/ImageRetrievalResearch: the timm model object
(`create_model`, `forward`, `forward_features`, `head`/`classifier`, timm state-dict keys) and the
cosine / top-k / ContrastiveLoss math around it, running as hand-written HIP kernels for gfx950
behind the C ABI in include/mi355_retrieval.h.  Host code is Python on PyTorch-ROCm; torch is used
for device memory, streams and torch.distributed only.
"""
from . import synth  # noqa: F401
from ._lib import MI355Error, lib, LIB_PATH  # noqa: F401
from . import models  # noqa: F401
from .rank import (ContrastiveLoss, CosineEmbeddingLoss, CosineSimilarity, Gallery, PreparedGallery, cos_sim_score_booster,  # noqa: F401
                   cos_sim_score_with_threshold, cosine_scores, cosine_topk,
                   distinct_class_topn, hit_counts, l2_normalize_rows, merge_topk, pair_cosine,
                   retrieval_metrics, synth_fill, topk, validation_metrics)

__all__ = ["create_model", "list_models", "load_checkpoint", "strip_lightning_prefix", "ContrastiveLoss", "CosineEmbeddingLoss", "validation_metrics", "CosineSimilarity", "Gallery", "PreparedGallery", "cosine_scores",
           "cosine_topk", "pair_cosine", "topk", "merge_topk", "hit_counts", "distinct_class_topn",
           "retrieval_metrics", "cos_sim_score_with_threshold", "cos_sim_score_booster", "l2_normalize_rows", "synth_fill", "ShardedGallery", "MI355Error"]


def __getattr__(name):  # lazy: models/sharded import torch.nn / torch.distributed
    if name in ("load_checkpoint", "strip_lightning_prefix"):
        from . import checkpoint
        return getattr(checkpoint, name)
    if name in ("create_model", "list_models", "ConvInput", "with_conv_input"):
        from . import models
        return getattr(models, name)
    if name == "ShardedGallery":
        from .sharded import ShardedGallery
        return ShardedGallery
    raise AttributeError(name)
