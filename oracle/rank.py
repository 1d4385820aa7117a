"""CPU oracle for the rank half of the hot path (TEST INFRASTRUCTURE — see oracle/__init__.py).

Reference call sites restated (all paths relative to /root/reference):
* all-pairs cosine + top-k ........ train/train.py:250-251, :345-356 (pinned semantics);
                                    notebook inference/training_analysis.ipynb raw :238 (k=150)
* pair cosine ..................... inference/inference.py:226,229; train/train.py:345-349
* ContrastiveLoss ................. utils/contrastive_loss.py:31-61
* hit counting .................... train/train.py:252-255 ; notebook raw :240-251 (3 distinct classes)
* score booster ................... utils/score_booster.py:1-37

Cosine formula: the one torch 2.10 implements:
    cos(x, y) = sum_d (x_d / max(||x||, eps)) * (y_d / max(||y||, eps)),  eps = 1e-6, fp32.
Tie rule for top-k (torch leaves it unspecified): higher score first, then LOWER index first.
"""
from __future__ import annotations

import numpy as np
import torch

EPS = 1e-6


# --------------------------------------------------------------------------- literal forms
def rank_reference_loop(Q: torch.Tensor, G: torch.Tensor, k: int, eps: float = EPS):
    """The reference's own call shape, one query at a time (train/train.py:250-251).

    for idx: sim = cos(fm_ims[idx].unsqueeze(0), fm_poss); vals, inds = torch.topk(sim, k)
    """
    cos = torch.nn.CosineSimilarity(dim=1, eps=eps)
    vals, inds = [], []
    for q in range(Q.shape[0]):
        sim = cos(Q[q].unsqueeze(0), G)
        v, i = torch.topk(sim, k=k)
        vals.append(v)
        inds.append(i)
    return torch.stack(vals), torch.stack(inds)


def pair_cosine_reference(A: torch.Tensor, B: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """inference/inference.py:226 — row-wise cos(A[i], B[i]) -> (B,)."""
    return torch.nn.CosineSimilarity(dim=1, eps=eps)(A, B)


# --------------------------------------------------------------------------- restatement
def l2_normalize_rows(X: np.ndarray, eps: float = EPS) -> np.ndarray:
    X = np.asarray(X, dtype=np.float32)
    n = np.sqrt((X.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    return (X / np.maximum(n, np.float32(eps))[:, None]).astype(np.float32)


def cosine_scores(Q: np.ndarray, G: np.ndarray, eps: float = EPS) -> np.ndarray:
    """(Q,D) x (G,D) -> (Q,G) fp32 cosine, normalise-then-dot (fp64 accumulate, rounded once)."""
    Qn = l2_normalize_rows(Q, eps).astype(np.float64)
    Gn = l2_normalize_rows(G, eps).astype(np.float64)
    return (Qn @ Gn.T).astype(np.float32)


def topk_rows(S: np.ndarray, k: int):
    """Per-row top-k of a score matrix: descending score, ties -> lower index first."""
    S = np.asarray(S, dtype=np.float32)
    Qn, G = S.shape
    k = min(k, G)
    # stable sort on -score keeps ascending index order among equal scores
    order = np.argsort(-S, axis=1, kind="stable")[:, :k]
    vals = np.take_along_axis(S, order, axis=1)
    return vals, order.astype(np.int64)


def rank_topk(Q: np.ndarray, G: np.ndarray, k: int, eps: float = EPS):
    return topk_rows(cosine_scores(Q, G, eps), k)


def kth_gap(S: np.ndarray, k: int) -> float:
    """Smallest gap between consecutive scores among the top-(k+1) of any row: a fixture is only
    index-certifiable if this is well above fp32 summation noise."""
    srt = -np.sort(-S.astype(np.float64), axis=1)[:, : k + 1]
    return float(np.min(srt[:, :-1] - srt[:, 1:]))


def pair_cosine(A: np.ndarray, B: np.ndarray, eps: float = EPS) -> np.ndarray:
    An = l2_normalize_rows(A, eps).astype(np.float64)
    Bn = l2_normalize_rows(B, eps).astype(np.float64)
    return (An * Bn).sum(1).astype(np.float32)


def contrastive_loss(fm1: np.ndarray, fm2: np.ndarray, label: float, margin: float,
                     mean: bool = True, eps: float = 1e-9) -> np.float32:
    """utils/contrastive_loss.py:56-61.

    dis = (fm2 - fm1).pow(2).sum(1)
    losses = 0.5 * (label * dis + (1 - label) * relu(margin - sqrt(dis + eps))^2)
    """
    d = (np.asarray(fm2, np.float32) - np.asarray(fm1, np.float32)).astype(np.float64)
    dis = (d * d).sum(1)
    hinge = np.maximum(margin - np.sqrt(dis + eps), 0.0)
    losses = 0.5 * (label * dis + (1.0 - label) * hinge * hinge)
    return np.float32(losses.mean() if mean else losses.sum())


def hit_counts(inds: np.ndarray, query_cls: np.ndarray, gallery_cls: np.ndarray):
    """train/train.py:252-255: top3 += cls[q] in cls[inds[:3]]; top1 += cls[q] == cls[inds[0]]."""
    c = gallery_cls[inds]
    top1 = int((c[:, 0] == query_cls).sum())
    top3 = int((c[:, :3] == query_cls[:, None]).any(1).sum())
    return top1, top3


def distinct_class_top3(inds: np.ndarray, vals: np.ndarray, gallery_cls: np.ndarray, n: int = 3):
    """Notebook raw :240-251: walk the ranked list, keep the first ``n`` DISTINCT classes.

    Returns (cls (Q,n), idx (Q,n), val (Q,n)); rows with fewer than n distinct classes are
    padded with -1 / -1 / nan (the notebook would simply produce a shorter list)."""
    Qn = inds.shape[0]
    oc = np.full((Qn, n), -1, np.int64)
    oi = np.full((Qn, n), -1, np.int64)
    ov = np.full((Qn, n), np.nan, np.float32)
    for q in range(Qn):
        seen = []
        for i, v in zip(inds[q], vals[q]):
            r = int(gallery_cls[int(i)])
            if r not in seen:
                oc[q, len(seen)] = r
                oi[q, len(seen)] = int(i)
                ov[q, len(seen)] = v
                seen.append(r)
            if len(seen) == n:
                break
    return oc, oi, ov


def score_boost(score: np.ndarray, eps: float, alpha: float, threshold: float = 0.0, mode: str = "threshold") -> np.ndarray:
    """utils/score_booster.py:17-20 (mode "threshold") and :33-36 ("for_pos" / "for_neg"), elementwise in fp32."""
    s = np.asarray(score, np.float32)
    e, a = np.float32(eps), np.float32(alpha)
    pos = (s + e) / (e + a)
    neg = np.abs((s + (a / e)) / (np.float32(2.0) * e))
    if mode == "for_pos":
        return pos.astype(np.float32)
    if mode == "for_neg":
        return neg.astype(np.float32)
    return np.where(s >= np.float32(threshold), pos, neg).astype(np.float32)
