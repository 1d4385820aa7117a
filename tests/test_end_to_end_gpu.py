"""The reference's whole inference loop on one small case (BASELINE.json configs[0]): EfficientNet-B3a embeds a
batch of query images and a batch of positives, then every query is ranked by cosine against a 1k-row gallery
(inference/inference.py:199-201 embed, :223-245 rank with torch.nn.CosineSimilarity(dim=1, eps=1e-6) + torch.topk).

Two checks:
  * rank parity on REAL embeddings: the GPU rank of the GPU embeddings equals the reference loop run on those same
    embeddings (indices bit-exact, scores within 1e-5);
  * whole-path closeness to the fp32 CPU path: the reference loop on the oracle's fp32 embeddings gives the same
    top-1 / top-3 wherever its own score gaps exceed the bf16 backbone's score error (measured and asserted below).
"""
import numpy as np
import pytest
import torch

import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet
from helpers import assert_topk_matches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NQ, G, K = 16, 1000, 3
TOL_SCORE_BF16 = 1.5e-2  # |cos(bf16 backbone) - cos(fp32 backbone)| over all (query, gallery) pairs: first order in the
                         # embeddings' ~0.5 % relative error for non-parallel pairs, second order for near-parallel ones


def images(seed, B, H=224, W=224):
    """Normalised-image-like inputs whose GLOBAL statistics differ per image (contrast x1.25 per step, per-channel
    offsets).  A random-init backbone followed by global average pooling maps same-statistics textures to embeddings
    that agree to 1e-5 in cosine — below bf16 noise — so texture alone cannot exercise a ranking."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    x = np.empty((B, 3, H, W), np.float32)
    for b in range(B):
        amp = 0.2 * 1.25 ** b
        for c in range(3):
            f = 2.0 + 3 * b + c
            x[b, c] = amp * np.sin(xx / f + b) * np.cos(yy / (f + 1.5) + c) + 1.5 * np.sin(1.7 * b + 2.1 * c)
    return (x + 0.1 * synth.normal(seed, x.shape)).astype(np.float32)


def reference_loop(fm_q, gallery, k):
    """inference.py:223-245 restated: one broadcast CosineSimilarity + topk per query (CPU, fp32)."""
    cos = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
    vals, inds = [], []
    for fm in fm_q:
        v, i = torch.topk(cos(fm, gallery), k=k)
        vals.append(v)
        inds.append(i)
    return torch.stack(vals).numpy(), torch.stack(inds).numpy()


def test_embed_then_rank_like_inference_py():
    sd = effnet.init_state_dict(5)
    model = M.create_model("efficientnet_b3a", num_classes=0).to(DEV).eval()
    model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classifier.")}, strict=True)

    q_img = images(21, NQ)
    p_img = (q_img + 0.1 * synth.normal(22, q_img.shape)).astype(np.float32)   # each positive = its query, perturbed
    xq, xp = torch.from_numpy(q_img), torch.from_numpy(p_img)

    # ---- fp32 CPU path (oracle = reference semantics; autocast is a no-op on CPU)
    fq32 = effnet.pool(effnet.forward_features(sd, xq, sim_bf16=False))
    fp32 = effnet.pool(effnet.forward_features(sd, xp, sim_bf16=False))
    # ---- HIP path: forward() with num_classes=0 returns the pooled embedding (timm contract)
    fq = model(xq.to(DEV))
    fp = model(xp.to(DEV))
    assert fq.shape == (NQ, 1536) and fq.dtype == torch.float32

    # gallery = the positives' embeddings followed by random rows of the same scale
    scale = float(fp32.norm(dim=1).mean()) / np.sqrt(1536.0)
    rnd = torch.from_numpy(synth.normal(23, (G - NQ, 1536))) * scale
    gal_gpu = torch.cat([fp, rnd.to(DEV)])
    gal_cpu = torch.cat([fp32, rnd])

    # (1) rank parity on the same (GPU) embeddings
    v, i = M.cosine_topk(fq, gal_gpu, K)
    v_ref, i_ref = reference_loop(fq.cpu(), gal_gpu.cpu(), K)
    s_all = torch.nn.functional.cosine_similarity(fq.cpu()[:, None, :], gal_gpu.cpu()[None], dim=2)
    top = torch.topk(s_all, K + 1).values
    gap = (top[:, :-1] - top[:, 1:]).min(1).values.numpy()
    ncert = assert_topk_matches(v.cpu(), i.cpu(), v_ref, i_ref, gap=gap, what="rank on HIP embeddings",
                                scores_ref=s_all.numpy())
    print(f"rank parity: {ncert} of {NQ} rows have certified gaps (> 1e-5); min gap {gap.min():.2e}")

    # (2) whole path vs the fp32 CPU path
    v32, i32 = reference_loop(fq32, gal_cpu, K)
    s32 = torch.nn.functional.cosine_similarity(fq32[:, None, :], gal_cpu[None], dim=2)
    err = float((s_all - s32).abs().max())
    print(f"scores: max |bf16 path - fp32 path| over all pairs = {err:.2e}")
    assert err < TOL_SCORE_BF16
    t32 = torch.topk(s32, K + 1).values.numpy()
    # per query: a gap larger than 2.5x the worst score error among that row's CONTENDERS (gallery rows within 0.05 of
    # its top-(K+1) boundary) cannot be closed by the backbone's error; rows far below cannot reach the top at all
    assert err < 0.02
    contend = s32 > torch.from_numpy(t32[:, K:K + 1]) - 0.05
    margins = 2.5 * ((s_all - s32).abs() * contend).max(1).values.numpy()
    top1_checked = top3_checked = 0
    for r in range(NQ):
        if t32[r, 0] - t32[r, 1] > margins[r]:
            assert int(i[r, 0]) == int(i32[r, 0]), (r, i[r].tolist(), i32[r].tolist())
            top1_checked += 1
        if t32[r, K - 1] - t32[r, K] > margins[r]:
            assert set(i[r].tolist()) == set(i32[r].tolist()), (r, i[r].tolist(), i32[r].tolist())
            top3_checked += 1
    print(f"top-1 compared on {top1_checked}, top-3 sets on {top3_checked} of {NQ} queries (margins {margins.min():.1e}..{margins.max():.1e})")
    # (a random-init backbone + GAP nearly collapses the embeddings, cos > 0.999 between most images, so only the
    #  queries with distinctive global statistics clear the margin; trained weights are not available offline)
    assert top1_checked >= 4, "test images too similar: the comparison would be vacuous"
    # the matching positive is row r of the gallery (inference.py:236-244 top-1 / top-3 accuracy)
    m = M.retrieval_metrics(fq, gal_gpu, torch.arange(NQ, device=DEV),
                            torch.cat([torch.arange(NQ), torch.full((G - NQ,), -1)]).to(DEV), k=K)
    top1_ref = float((i32[:, 0] == np.arange(NQ)).mean())
    top3_ref = float((i32 == np.arange(NQ)[:, None]).any(1).mean())
    print(f"top-1 accuracy {m['top1']:.3f} (fp32 path {top1_ref:.3f}), top-3 {m['top3']:.3f} ({top3_ref:.3f})")
    slack = (NQ - top1_checked) / NQ + 1e-9
    assert abs(m["top1"] - top1_ref) <= slack and abs(m["top3"] - top3_ref) <= (NQ - top3_checked) / NQ + 1e-9
