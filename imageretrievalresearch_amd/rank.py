"""Host side of the rank half: cosine similarity, top-k, ContrastiveLoss, retrieval metrics.

Mirrors the torch objects the reference uses (same names, argument meaning, error behaviour) and
routes the arithmetic to libmi355_retrieval:

* ``CosineSimilarity(dim=1, eps=1e-6)`` ....... train/train.py:73, inference/inference.py:169
* ``topk(sim, k)`` ............................ train/train.py:251,356 ; notebook raw :238
* ``cosine_topk(Q, G, k)`` .................... the whole per-query loop train/train.py:249-255 as one call
* ``ContrastiveLoss(margin)(fm1, fm2, label, mean)`` ... utils/contrastive_loss.py:6-61
* ``hit_counts`` / ``distinct_class_topn`` .... train/train.py:252-255 ; notebook raw :240-251
"""
from __future__ import annotations

import torch

from ._lib import MI355Error, check, lib, require_cuda, stream_ptr

_EPS = 1e-6


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    require_cuda(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _Workspace:
    """Scratch that only grows, one buffer per (device, stream): calls issued on different streams (bench.py runs the rank
    beside the embed) never share a workspace, and a buffer that is replaced by a bigger one is handed back to torch's
    caching allocator on the stream that used it (``record_stream``), so it is not recycled under a running kernel."""

    def __init__(self):
        self.buf = {}

    def get(self, device, nbytes: int) -> torch.Tensor:
        device = torch.device(device)
        stream = torch.cuda.current_stream(device)
        key = (device, stream.cuda_stream)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None:
                b.record_stream(stream)
            with torch.cuda.stream(stream):
                b = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_ws = _Workspace()


def synth_fill(n: int, seed: int, kind: int, device, offset: int = 0) -> torch.Tensor:
    """Device-side portable generator (bit-identical to ``synth.fill``)."""
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    require_cuda(out, "synth_fill output")
    with torch.cuda.device(out.device):
        check(lib().mi355_synth_fill(out.data_ptr(), int(n), int(seed), int(offset), int(kind),
                                     stream_ptr(out.device)))
    return out


def l2_normalize_rows(x: torch.Tensor, eps: float = _EPS, out: torch.Tensor | None = None) -> torch.Tensor:
    x = _f32c(x, "x")
    if x.dim() != 2:
        raise MI355Error(f"l2_normalize_rows expects (rows, dim), got {tuple(x.shape)}")
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().mi355_l2_normalize_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], eps,
                                            stream_ptr(x.device)))
    return out


def _check_qg(q, g):
    if q.dim() != 2 or g.dim() != 2:
        raise MI355Error(f"expected (Q,D) queries and (G,D) gallery, got {tuple(q.shape)} and {tuple(g.shape)}")
    if q.shape[1] != g.shape[1]:
        raise MI355Error(f"embedding dims differ: {q.shape[1]} vs {g.shape[1]}")
    if q.device != g.device:
        raise MI355Error(f"queries on {q.device} but gallery on {g.device}")


def cosine_scores(queries: torch.Tensor, gallery: torch.Tensor, eps: float = _EPS,
                  gallery_is_normalized: bool = False) -> torch.Tensor:
    """(Q,D) x (G,D) -> (Q,G) fp32 cosine matrix."""
    q, g = _f32c(queries, "queries"), _f32c(gallery, "gallery")
    _check_qg(q, g)
    Q, D = q.shape
    G = g.shape[0]
    out = torch.empty((Q, G), dtype=torch.float32, device=q.device)
    if Q == 0 or G == 0:
        return out
    nbytes = lib().mi355_rank_workspace_bytes(Q, G, D, 0)
    ws = _ws.get(q.device, nbytes)
    with torch.cuda.device(q.device):
        check(lib().mi355_cosine_scores(q.data_ptr(), Q, g.data_ptr(), G, D, int(gallery_is_normalized), eps,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(q.device)))
    return out


def cosine_topk(queries: torch.Tensor, gallery: torch.Tensor, k: int, eps: float = _EPS,
                gallery_is_normalized: bool = False, idx_offset: int = 0):
    """All-pairs cosine + top-k: the loop of train/train.py:249-251 as one call.

    Returns (values (Q,k) fp32, indices (Q,k) int64), sorted by descending score; equal scores are
    ordered by ascending gallery index.  Raises like ``torch.topk`` when k exceeds the gallery size."""
    q, g = _f32c(queries, "queries"), _f32c(gallery, "gallery")
    _check_qg(q, g)
    Q, D = q.shape
    G = g.shape[0]
    if k > G or k < 1:
        raise MI355Error(f"selected index k out of range: k={k}, gallery rows={G}")
    vals = torch.empty((Q, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
    if Q == 0:
        return vals, idx
    nbytes = lib().mi355_rank_workspace_bytes(Q, G, D, k)
    ws = _ws.get(q.device, nbytes)
    with torch.cuda.device(q.device):
        check(lib().mi355_rank_topk(q.data_ptr(), Q, g.data_ptr(), G, D, int(gallery_is_normalized), k, eps,
                                    int(idx_offset), vals.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                    stream_ptr(q.device)))
    return vals, idx


class PreparedGallery:
    """The three bf16 planes of a gallery's NORMALISED rows in the cosine GEMM's fragment order (6 B per element): made once
    for a resident gallery (``mi355_gallery_prepare``), so that a search does no per-call work on the gallery side - the
    reference re-reads and re-normalises the whole gallery for every query (train/train.py:250)."""

    def __init__(self, gallery_normalized: torch.Tensor):
        g = _f32c(gallery_normalized, "gallery")
        if g.dim() != 2 or g.shape[0] < 1:
            raise MI355Error(f"a prepared gallery needs (G, D) rows with G >= 1, got {tuple(g.shape)}")
        self.rows, self.dim, self.device = int(g.shape[0]), int(g.shape[1]), g.device
        nbytes = lib().mi355_gallery_planes_bytes(self.rows, self.dim)
        self.planes = torch.empty((nbytes,), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            check(lib().mi355_gallery_prepare(g.data_ptr(), self.rows, self.dim, self.planes.data_ptr(), nbytes,
                                              stream_ptr(g.device)))

    @staticmethod
    def supports(Q: int, k: int) -> bool:
        """The prepared path covers the fused selection's range; other shapes use the fp32 rows (``cosine_topk``)."""
        return Q > 4 and 1 <= k <= 8

    def search(self, queries: torch.Tensor, k: int, eps: float = _EPS, idx_offset: int = 0):
        """``cosine_topk(queries, rows, k, gallery_is_normalized=True)`` - same values and indices, bit for bit."""
        q = _f32c(queries, "queries")
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise MI355Error(f"queries must be (Q,{self.dim}), got {tuple(q.shape)}")
        Q = q.shape[0]
        if k > self.rows or k < 1:
            raise MI355Error(f"selected index k out of range: k={k}, gallery rows={self.rows}")
        if not self.supports(Q, k):
            raise MI355Error(f"prepared search needs k <= 8 and more than 4 queries (got k={k}, Q={Q}): use cosine_topk on the rows")
        vals = torch.empty((Q, k), dtype=torch.float32, device=q.device)
        idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
        ws = _ws.get(q.device, lib().mi355_rank_workspace_bytes(Q, self.rows, self.dim, k))
        with torch.cuda.device(q.device):
            check(lib().mi355_rank_topk_prepared(q.data_ptr(), Q, self.planes.data_ptr(), self.rows, self.dim, k, eps,
                                                 int(idx_offset), vals.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 stream_ptr(q.device)))
        return vals, idx


def topk(scores: torch.Tensor, k: int, idx_offset: int = 0):
    """``torch.topk(sim, k)`` over the last dim of a 1-D or 2-D fp32 score tensor."""
    squeeze = scores.dim() == 1
    s = _f32c(scores.unsqueeze(0) if squeeze else scores, "scores")
    if s.dim() != 2:
        raise MI355Error(f"topk expects a 1-D or 2-D tensor, got {tuple(scores.shape)}")
    Q, G = s.shape
    if k > G or k < 1:
        raise MI355Error(f"selected index k out of range: k={k}, row length={G}")
    vals = torch.empty((Q, k), dtype=torch.float32, device=s.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=s.device)
    if Q:
        nbytes = lib().mi355_rank_workspace_bytes(Q, G, 0, k)
        ws = _ws.get(s.device, nbytes)
        with torch.cuda.device(s.device):
            check(lib().mi355_topk_rows(s.data_ptr(), Q, G, k, int(idx_offset), vals.data_ptr(), idx.data_ptr(),
                                        ws.data_ptr(), ws.numel(), stream_ptr(s.device)))
    return (vals[0], idx[0]) if squeeze else (vals, idx)


def merge_topk(cand_val: torch.Tensor, cand_idx: torch.Tensor, k: int):
    """Merge (Q, ncand) candidate lists (e.g. all-gathered per-shard top-k) into the global top-k."""
    v = _f32c(cand_val, "cand_val")
    require_cuda(cand_idx, "cand_idx")
    i = cand_idx.to(torch.int64).contiguous()
    if v.shape != i.shape or v.dim() != 2:
        raise MI355Error(f"merge_topk expects matching (Q, ncand) tensors, got {tuple(v.shape)} / {tuple(i.shape)}")
    Q, n = v.shape
    if k > n or k < 1:
        raise MI355Error(f"selected index k out of range: k={k}, candidates={n}")
    vals = torch.empty((Q, k), dtype=torch.float32, device=v.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=v.device)
    if Q:
        nbytes = lib().mi355_rank_workspace_bytes(Q, n, 0, k)
        ws = _ws.get(v.device, nbytes)
        with torch.cuda.device(v.device):
            check(lib().mi355_merge_topk(v.data_ptr(), i.data_ptr(), Q, n, k, vals.data_ptr(), idx.data_ptr(),
                                         ws.data_ptr(), ws.numel(), stream_ptr(v.device)))
    return vals, idx


def pack_candidates(vals: "torch.Tensor | None", idx: "torch.Tensor | None", Q: int, k: int, device) -> torch.Tensor:
    """(Q, kk) local top-k results (kk <= k; None for an empty shard) -> (Q, k, 2) int32 {score bits, LOCAL row index},
    missing slots = {-inf, -1}: the one tensor a rank contributes to the sharded search's candidate all-gather."""
    packed = torch.empty((Q, k, 2), dtype=torch.int32, device=device)
    kk = 0 if vals is None else vals.shape[1]
    if Q:
        v = _f32c(vals, "vals") if kk else None
        i = idx.to(torch.int64).contiguous() if kk else None
        with torch.cuda.device(packed.device):
            check(lib().mi355_pack_candidates(v.data_ptr() if kk else None, i.data_ptr() if kk else None, Q, kk, k,
                                              packed.data_ptr(), stream_ptr(packed.device)))
    return packed


def merge_packed_topk(packed: torch.Tensor, shard_offsets: torch.Tensor, k: int):
    """All-gathered packed candidates (world, Q, k, 2) int32 + the shards' first global rows (world,) int64 on the device
    -> global top-k (values (Q, k) f32, indices (Q, k) int64): offsets, unpacking and the merge run in the library."""
    require_cuda(packed, "packed candidates")
    if packed.dtype != torch.int32 or packed.dim() != 4 or packed.shape[3] != 2 or packed.shape[2] != k:
        raise MI355Error(f"merge_packed_topk expects (world, Q, {k}, 2) int32, got {packed.dtype} {tuple(packed.shape)}")
    packed = packed.contiguous()
    world, Q = packed.shape[0], packed.shape[1]
    off = shard_offsets.to(packed.device, torch.int64).contiguous().view(-1)
    if off.numel() != world:
        raise MI355Error(f"merge_packed_topk: {off.numel()} shard offsets for {world} shards")
    vals = torch.empty((Q, k), dtype=torch.float32, device=packed.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=packed.device)
    if Q:
        ws = _ws.get(packed.device, lib().mi355_merge_packed_workspace_bytes(Q, world, k))
        with torch.cuda.device(packed.device):
            check(lib().mi355_merge_packed_topk(packed.data_ptr(), off.data_ptr(), world, Q, k, vals.data_ptr(),
                                                idx.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(packed.device)))
    return vals, idx


def pair_cosine(a: torch.Tensor, b: torch.Tensor, eps: float = _EPS) -> torch.Tensor:
    """Row-wise cos(a[i], b[i]) — inference/inference.py:226."""
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if a.shape != b.shape or a.dim() != 2:
        raise MI355Error(f"pair_cosine expects two (B,D) tensors, got {tuple(a.shape)} / {tuple(b.shape)}")
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    if a.shape[0]:
        with torch.cuda.device(a.device):
            check(lib().mi355_pair_cosine(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], eps, out.data_ptr(),
                                          stream_ptr(a.device)))
    return out


class CosineSimilarity(torch.nn.Module):
    """``torch.nn.CosineSimilarity(dim=1, eps)`` for the two call shapes on the hot path:
    (1,D) vs (G,D) -> (G,)   [train/train.py:250]   and   (B,D) vs (B,D) -> (B,)   [inference.py:226]."""

    def __init__(self, dim: int = 1, eps: float = _EPS):
        super().__init__()
        if dim != 1:
            raise MI355Error("only dim=1 (the reference's setting) is implemented")
        self.dim, self.eps = dim, eps

    def forward(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        if x1.dim() != 2 or x2.dim() != 2:
            raise MI355Error(f"CosineSimilarity expects 2-D inputs, got {tuple(x1.shape)} / {tuple(x2.shape)}")
        if x1.shape == x2.shape and x1.shape[0] != 1:
            return pair_cosine(x1, x2, self.eps)
        if x1.shape[0] == 1:
            return cosine_scores(x1, x2, self.eps)[0]
        if x2.shape[0] == 1:
            return cosine_scores(x2, x1, self.eps)[0]
        raise MI355Error(f"unsupported broadcast {tuple(x1.shape)} vs {tuple(x2.shape)}")


class ContrastiveLoss(torch.nn.Module):
    """utils/contrastive_loss.py:6-61, forward only (the inference loop calls it under no_grad,
    inference/inference.py:193-204)."""

    def __init__(self, margin):
        super().__init__()
        self.margin, self.eps = margin, 1e-9

    def forward(self, fm1, fm2, label, mean=True):
        a, b = _f32c(fm1, "fm1"), _f32c(fm2, "fm2")
        if a.shape != b.shape or a.dim() != 2:
            raise MI355Error(f"ContrastiveLoss expects two (B,D) tensors, got {tuple(a.shape)} / {tuple(b.shape)}")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib().mi355_contrastive_loss(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], float(label),
                                               float(self.margin), int(bool(mean)), out.data_ptr(), None,
                                               stream_ptr(a.device)))
        return out


class CosineEmbeddingLoss(torch.nn.Module):
    """``torch.nn.CosineEmbeddingLoss(margin)`` for the reference's call shape: two (B,D) batches and a scalar
    target of +1 or -1 (``labels["pos"]`` / ``labels["neg"]``, train/train.py:81, :214-216).  Forward only."""

    def __init__(self, margin: float = 0.0, reduction: str = "mean"):
        super().__init__()
        if reduction not in ("mean", "sum"):
            raise MI355Error("reduction must be 'mean' or 'sum'")
        self.margin, self.reduction = margin, reduction

    def forward(self, input1, input2, target):
        a, b = _f32c(input1, "input1"), _f32c(input2, "input2")
        if a.shape != b.shape or a.dim() != 2:
            raise MI355Error(f"CosineEmbeddingLoss expects two (B,D) tensors, got {tuple(a.shape)} / {tuple(b.shape)}")
        t = float(target.reshape(-1)[0].item()) if torch.is_tensor(target) else float(target)
        if torch.is_tensor(target) and target.numel() != 1:
            raise MI355Error("only a scalar (broadcast) target is supported, as in the reference")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib().mi355_cosine_embedding_loss(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], t,
                                                    float(self.margin), int(self.reduction == "mean"), out.data_ptr(),
                                                    stream_ptr(a.device)))
        return out


def validation_metrics(fm_ims, fm_poss, fm_negs, clss, margin: float = 0.5, k: int = 3):
    """The metric half of ``validation_step`` (train/train.py:308-373) as a handful of batched launches instead of a
    Python loop with one cosine + topk + ``.item()`` per row: cosine-embedding losses (+1 / -1 targets), mean pair
    cosines (``cos_sims`` / ``cos_unsims``), and top-1 / top-3 of every query against the positives of the batch.
    Values stay on the device; nothing here synchronises."""
    cel = CosineEmbeddingLoss(margin)
    loss_pos = cel(fm_ims, fm_poss, 1.0)
    loss_neg = cel(fm_ims, fm_negs, -1.0)
    vals, inds = cosine_topk(fm_ims, fm_poss, min(k, fm_poss.shape[0]))
    counts = hit_counts(inds, clss, clss)
    n = fm_ims.shape[0]
    return {"loss_cos_poss": loss_pos, "loss_cos_negs": loss_neg, "loss_cos": loss_pos + loss_neg,
            "cos_sims": pair_cosine(fm_ims, fm_poss).mean(), "cos_unsims": pair_cosine(fm_ims, fm_negs).mean(),
            "top1": counts[0].float() / n, "top3": counts[1].float() / n, "topk_vals": vals, "topk_inds": inds}


def hit_counts(idx: torch.Tensor, query_cls: torch.Tensor, gallery_cls: torch.Tensor):
    """train/train.py:252-255 -> (top1_hits, top3_hits) as a device int64[2] tensor (no sync)."""
    require_cuda(idx, "idx")
    idx = idx.to(torch.int64).contiguous()
    qc = query_cls.to(idx.device, torch.int64).contiguous()
    gc = gallery_cls.to(idx.device, torch.int64).contiguous()
    if idx.dim() != 2 or qc.shape[0] != idx.shape[0]:
        raise MI355Error("hit_counts expects idx (Q,k) and query_cls (Q,)")
    counts = torch.zeros(2, dtype=torch.int64, device=idx.device)
    if idx.shape[0]:
        with torch.cuda.device(idx.device):
            check(lib().mi355_hit_counts(idx.data_ptr(), idx.shape[0], idx.shape[1], qc.data_ptr(), gc.data_ptr(),
                                         gc.numel(), counts.data_ptr(), stream_ptr(idx.device)))
    return counts


def distinct_class_topn(idx: torch.Tensor, val: torch.Tensor, gallery_cls: torch.Tensor, n: int = 3):
    """Notebook raw :240-251: first n distinct classes along each ranked list."""
    require_cuda(idx, "idx")
    idx = idx.to(torch.int64).contiguous()
    val = _f32c(val, "val")
    gc = gallery_cls.to(idx.device, torch.int64).contiguous()
    Q, k = idx.shape
    oc = torch.empty((Q, n), dtype=torch.int64, device=idx.device)
    oi = torch.empty((Q, n), dtype=torch.int64, device=idx.device)
    ov = torch.empty((Q, n), dtype=torch.float32, device=idx.device)
    if Q:
        with torch.cuda.device(idx.device):
            check(lib().mi355_distinct_class_topn(idx.data_ptr(), val.data_ptr(), Q, k, gc.data_ptr(), gc.numel(), n,
                                                  oc.data_ptr(), oi.data_ptr(), ov.data_ptr(),
                                                  stream_ptr(idx.device)))
    return oc, oi, ov


def _score_boost(score, eps, alpha, threshold, mode):
    s = _f32c(score, "score")
    out = torch.empty_like(s)
    if s.numel():
        with torch.cuda.device(s.device):
            check(lib().mi355_score_boost(s.data_ptr(), s.numel(), float(eps), float(alpha), float(threshold), mode,
                                          out.data_ptr(), stream_ptr(s.device)))
    return out


def cos_sim_score_with_threshold(score: torch.Tensor, eps: float, alpha: float, threshold: float) -> torch.Tensor:
    """utils/score_booster.py:1-20 over a whole fp32 score tensor (the reference takes one score at a time and
    prints it; the print is not reproduced)."""
    return _score_boost(score, eps, alpha, threshold, 0)


def cos_sim_score_booster(score: torch.Tensor, eps: float, alpha: float, mode: str) -> torch.Tensor:
    """utils/score_booster.py:22-37; ``mode`` is "for_pos" or "for_neg" (anything else returns None there; here it raises)."""
    if mode not in ("for_pos", "for_neg"):
        raise MI355Error(f'cos_sim_score_booster: mode must be "for_pos" or "for_neg", got {mode!r}')
    return _score_boost(score, eps, alpha, 0.0, 1 if mode == "for_pos" else 2)


class Gallery:
    """Resident gallery: rows are L2-normalised once when added and stay in HBM as fp32 (SURVEY §8e:
    the gallery is *born* on the GPU that embedded it).  ``search`` is then one fused call per query
    batch instead of the reference's per-query cosine + topk pair."""

    def __init__(self, dim: int, device, capacity: int = 0, eps: float = _EPS):
        self.dim, self.device, self.eps = int(dim), torch.device(device), eps
        self.rows = 0
        self._buf = torch.empty((max(capacity, 0), self.dim), dtype=torch.float32, device=self.device)
        self.labels = None

    def _reserve(self, n):
        if n > self._buf.shape[0]:
            cap = max(n, int(self._buf.shape[0] * 1.5) + 1024)
            nb = torch.empty((cap, self.dim), dtype=torch.float32, device=self.device)
            nb[: self.rows].copy_(self._buf[: self.rows])
            self._buf = nb

    def add(self, embeddings: torch.Tensor, labels: torch.Tensor | None = None):
        e = _f32c(embeddings, "embeddings")
        if e.dim() != 2 or e.shape[1] != self.dim:
            raise MI355Error(f"gallery rows must be (n,{self.dim}), got {tuple(e.shape)}")
        n = e.shape[0]
        self._reserve(self.rows + n)
        if n:
            l2_normalize_rows(e, self.eps, out=self._buf[self.rows: self.rows + n])
        if labels is not None:
            lab = labels.to(self.device, torch.int64)
            self.labels = lab if self.labels is None else torch.cat([self.labels, lab])
        self.rows += n
        return self

    @property
    def data(self) -> torch.Tensor:
        return self._buf[: self.rows]

    def __len__(self):
        return self.rows

    def prepare(self):
        """Split the resident rows once into the GEMM's bf16 planes (PreparedGallery, +6 B per element): searches with k <= 8
        and more than 4 queries then run without touching the fp32 rows.  Call again after ``add``."""
        self._prepared = PreparedGallery(self.data) if self.rows else None
        self._prepared_rows = self.rows
        return self

    def search(self, queries: torch.Tensor, k: int, idx_offset: int = 0):
        p = getattr(self, "_prepared", None)
        if p is not None and self._prepared_rows == self.rows and PreparedGallery.supports(queries.shape[0], k):
            return p.search(queries, k, self.eps, idx_offset)
        return cosine_topk(queries, self.data, k, self.eps, gallery_is_normalized=True, idx_offset=idx_offset)


def retrieval_metrics(queries, positives, query_cls, gallery_cls=None, k: int = 3):
    """The rank loop of inference/inference.py:223-245 with the pinned semantics of train/train.py:
    mean pair cosine, top-1 and top-3 accuracy of ``queries`` against the ``positives`` gallery."""
    gallery_cls = query_cls if gallery_cls is None else gallery_cls
    vals, idx = cosine_topk(queries, positives, k)
    counts = hit_counts(idx, query_cls, gallery_cls)
    pos = pair_cosine(queries, positives) if queries.shape == positives.shape else None
    n = queries.shape[0]
    return {"top1": counts[0].item() / n, "top3": counts[1].item() / n,
            "scores": None if pos is None else pos.mean().item(), "topk_vals": vals, "topk_inds": idx}
