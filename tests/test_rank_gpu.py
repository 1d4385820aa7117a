"""Parity of the HIP rank path (through the C ABI) against the oracle and the reference-generated goldens."""
import numpy as np
import pytest
import torch

from helpers import SCORE_TOL, assert_topk_matches, load_golden
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import rank as orank

pytestmark = pytest.mark.gpu
GOLD = load_golden()
DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def test_native_library_is_loaded():
    assert M.lib().mi355_device_count() >= 1


@pytest.mark.parametrize("kind", [synth.UNIFORM, synth.NORMAL])
def test_synth_fill_bit_identical_to_numpy(kind):
    n, seed, off = 100003, 77, 12345
    got = M.synth_fill(n, seed, kind, DEV, offset=off).cpu().numpy()
    np.testing.assert_array_equal(got, synth.fill(seed, n, kind, offset=off))


@pytest.mark.parametrize("case", ["cfg1", "cfg2", "g100k"])
def test_cosine_topk_matches_reference_goldens(case):
    qs, Qn, gs, Gn, d = (int(x) for x in GOLD[f"{case}_meta"])
    Q = M.synth_fill(Qn * d, qs, synth.NORMAL, DEV).view(Qn, d)
    G = M.synth_fill(Gn * d, gs, synth.NORMAL, DEV).view(Gn, d)
    ks = [k for k in (1, 3, 150) if f"{case}_k{k}_idx" in GOLD.files]
    for pre in (False, True):
        gal = M.l2_normalize_rows(G) if pre else G
        for k in ks:
            v, i = M.cosine_topk(Q, gal, k, gallery_is_normalized=pre)
            ncert = assert_topk_matches(v.cpu().numpy(), i.cpu().numpy(), GOLD[f"{case}_k{k}_val"],
                                        GOLD[f"{case}_k{k}_idx"], GOLD[f"{case}_k{k}_gap"],
                                        float(GOLD["cert_gap"]), f"{case} k={k} pre={pre}")
            if k <= 3:
                assert ncert >= Qn - 2


def test_1m_gallery_probe_queries():
    if "g1m_meta" not in GOLD.files:
        pytest.skip("1M-row goldens not generated")
    qs, Qn, gs, Gn, d = (int(x) for x in GOLD["g1m_meta"])
    Q = M.synth_fill(Qn * d, qs, synth.NORMAL, DEV).view(Qn, d)
    G = M.synth_fill(Gn * d, gs, synth.NORMAL, DEV).view(Gn, d)
    v, i = M.cosine_topk(Q, G, 3)
    assert_topk_matches(v.cpu().numpy(), i.cpu().numpy(), GOLD["g1m_k3_val"], GOLD["g1m_k3_idx"],
                        GOLD["g1m_k3_gap"], float(GOLD["cert_gap"]), "g1m k=3")


@pytest.mark.parametrize("Q,G,D,k", [(1, 1, 4, 1), (1, 7, 5, 3), (3, 129, 33, 8), (65, 300, 100, 5),
                                      (130, 2049, 64, 9), (257, 1000, 1000, 150), (16, 5000, 1920, 1024),
                                      (2, 20000, 17, 4)])
def test_ragged_shapes_against_oracle(Q, G, D, k):
    q, g = synth.normal(100 + Q, (Q, D)), synth.normal(200 + G, (G, D))
    want_v, want_i = orank.rank_topk(q, g, k)
    S = orank.cosine_scores(q, g)
    gaps = np.array([orank.kth_gap(S[r:r + 1], min(k, G - 1)) if G > 1 else 1.0 for r in range(Q)])
    v, i = M.cosine_topk(dev(q), dev(g), k)
    assert_topk_matches(v.cpu().numpy(), i.cpu().numpy(), want_v, want_i, gaps, 1e-5, f"Q{Q} G{G} D{D} k{k}")
    s = M.cosine_scores(dev(q), dev(g)).cpu().numpy()
    np.testing.assert_allclose(s, S, atol=SCORE_TOL)


def test_ties_resolve_to_lower_index():
    S = np.zeros((3, 5000), np.float32)
    S[0, [10, 4000, 77]] = 1.0          # three-way tie at the top
    S[1, :] = 0.25                      # everything tied
    S[2, 4999] = 2.0
    for k in (1, 3, 8, 40):
        v, i = M.topk(dev(S), k)
        wv, wi = orank.topk_rows(S, k)
        np.testing.assert_array_equal(i.cpu().numpy(), wi)
        np.testing.assert_array_equal(v.cpu().numpy(), wv)
    # duplicate gallery rows tie exactly in the fused path as well
    g = synth.normal(5, (300, 64))
    g[250] = g[3]
    g[17] = g[3]
    v, i = M.cosine_topk(dev(g[3:4]), dev(g), 3)
    assert i.cpu().numpy().tolist() == [[3, 17, 250]]


def test_topk_1d_like_torch_and_errors():
    s = synth.normal(9, (1000,))
    v, i = M.topk(dev(s), 3)
    tv, ti = torch.topk(torch.from_numpy(s), 3)
    assert i.cpu().tolist() == ti.tolist() and torch.equal(v.cpu(), tv)
    with pytest.raises(M.MI355Error):
        M.topk(dev(s), 1001)
    with pytest.raises(M.MI355Error):
        M.cosine_topk(dev(synth.normal(1, (2, 8))), dev(synth.normal(2, (3, 8))), 4)
    with pytest.raises(M.MI355Error):
        M.cosine_topk(dev(synth.normal(1, (2, 8))), dev(synth.normal(2, (3, 9))), 1)
    with pytest.raises(M.MI355Error):
        M.cosine_topk(torch.zeros(2, 8), torch.zeros(3, 8), 1)   # CPU tensors: no fallback


def test_merge_topk_equals_unsharded():
    q, g = synth.normal(31, (37, 256)), synth.normal(32, (4000, 256))
    k = 5
    full_v, full_i = M.cosine_topk(dev(q), dev(g), k)
    cv, ci = [], []
    bounds = [0, 1000, 1001, 2500, 4000]       # ragged shards, one of a single row... (k > rows handled below)
    for a, b in zip(bounds[:-1], bounds[1:]):
        kk = min(k, b - a)
        v, i = M.cosine_topk(dev(q), dev(g[a:b]), kk, idx_offset=a)
        if kk < k:  # pad short shards the way ShardedGallery does
            v = torch.cat([v, torch.full((37, k - kk), -float("inf"), device=DEV)], 1)
            i = torch.cat([i, torch.full((37, k - kk), 2 ** 62, dtype=torch.int64, device=DEV)], 1)
        cv.append(v)
        ci.append(i)
    mv, mi = M.merge_topk(torch.cat(cv, 1), torch.cat(ci, 1), k)
    assert torch.equal(mi, full_i) and torch.equal(mv, full_v)


def test_pair_cosine_and_module_shapes():
    for case in ("cfg1", "cfg2"):
        qs, Qn, gs, Gn, d = (int(x) for x in GOLD[f"{case}_meta"])
        Q, G = synth.normal(qs, (Qn, d)), synth.normal(gs, (Qn, d))
        got = M.pair_cosine(dev(Q), dev(G)).cpu().numpy()
        np.testing.assert_allclose(got, GOLD[f"{case}_pair"], atol=SCORE_TOL)
    cos = M.CosineSimilarity(dim=1, eps=1e-6)
    a, b = synth.normal(1, (6, 40)), synth.normal(2, (50, 40))
    one_vs_many = cos(dev(a[:1]), dev(b)).cpu().numpy()          # train/train.py:250 shape
    np.testing.assert_allclose(one_vs_many, orank.cosine_scores(a[:1], b)[0], atol=SCORE_TOL)
    z = np.zeros((2, 40), np.float32)                            # zero rows: eps clamp, no NaN
    assert torch.isfinite(cos(dev(z), dev(a[:2]))).all()


def test_contrastive_loss_matches_reference_goldens():
    s1, s2, n, d = (int(x) for x in GOLD["cl_seeds"])
    a = dev(synth.normal(s1, (n, d)) * np.float32(GOLD["cl_scale"]))
    b = dev(synth.normal(s2, (n, d)) * np.float32(GOLD["cl_scale"]))
    for margin, label, mean, want in GOLD["cl_cases"]:
        got = M.ContrastiveLoss(margin)(a, b, label, bool(mean)).item()
        assert got == pytest.approx(want, rel=1e-5), (margin, label, mean)
    s1, s2, n, d = (int(x) for x in GOLD["cl_infer_seeds"])
    a, b = dev(synth.normal(s1, (n, d))), dev(synth.normal(s2, (n, d)))
    assert M.ContrastiveLoss(0.5)(a, b, 1.).item() == pytest.approx(GOLD["cl_infer"][0], rel=1e-5)
    assert M.ContrastiveLoss(0.5)(a, b, 0., False).item() == pytest.approx(GOLD["cl_infer"][1], rel=1e-5, abs=1e-7)


def test_hit_counts_and_distinct_classes_match_oracle():
    q, g = synth.normal(41, (200, 128)), synth.normal(42, (3000, 128))
    gcls = (np.arange(3000) * 7919 % 25).astype(np.int64)
    qcls = (np.arange(200) * 31 % 25).astype(np.int64)
    v, i = M.cosine_topk(dev(q), dev(g), 150)
    want = orank.hit_counts(i.cpu().numpy(), qcls, gcls)
    got = M.hit_counts(i, dev(qcls), dev(gcls)).cpu().tolist()
    assert tuple(got) == want
    oc, oi, ov = M.distinct_class_topn(i, v, dev(gcls), 3)
    wc, wi, wv = orank.distinct_class_top3(i.cpu().numpy(), v.cpu().numpy(), gcls)
    np.testing.assert_array_equal(oc.cpu().numpy(), wc)
    np.testing.assert_array_equal(oi.cpu().numpy(), wi)
    np.testing.assert_array_equal(ov.cpu().numpy(), wv)


def test_rank_is_deterministic_run_to_run():
    Q = M.synth_fill(256 * 1536, 1, synth.NORMAL, DEV).view(256, 1536)
    G = M.synth_fill(20000 * 1536, 2, synth.NORMAL, DEV).view(20000, 1536)
    v1, i1 = M.cosine_topk(Q, G, 3)
    v2, i2 = M.cosine_topk(Q, G, 3)
    assert torch.equal(v1, v2) and torch.equal(i1, i2)


def test_gallery_object_and_metrics():
    emb = synth.normal(51, (500, 96))
    cls = (np.arange(500) % 10).astype(np.int64)
    gal = M.Gallery(96, DEV)
    gal.add(dev(emb[:200]), dev(cls[:200])).add(dev(emb[200:]), dev(cls[200:]))
    assert len(gal) == 500
    v, i = gal.search(dev(emb[:64]), 3)
    assert (i[:, 0].cpu().numpy() == np.arange(64)).all()           # each row finds itself first
    np.testing.assert_allclose(v[:, 0].cpu().numpy(), 1.0, atol=SCORE_TOL)
    m = M.retrieval_metrics(dev(emb[:64]), dev(emb[:64]), dev(cls[:64]))
    assert m["top1"] == 1.0 and m["top3"] == 1.0 and abs(m["scores"] - 1.0) < SCORE_TOL


def test_cosine_embedding_loss_and_validation_metrics():
    """f-4: train/train.py:308-373 — against torch's own CosineEmbeddingLoss / per-row loop on the CPU."""
    q, p, n = synth.normal(61, (48, 1536)), synth.normal(62, (48, 1536)), synth.normal(63, (48, 1536))
    p = (0.7 * q + 0.3 * p).astype(np.float32)                          # positives correlate with their query
    cls = (np.arange(48) % 12).astype(np.int64)
    tq, tp, tn = (torch.from_numpy(x) for x in (q, p, n))
    for margin in (0.0, 0.5):
        ref = torch.nn.CosineEmbeddingLoss(margin=margin)
        for tgt, other in ((1.0, tp), (-1.0, tn)):
            want = ref(tq, other, torch.tensor(tgt).unsqueeze(0)).item()      # labels["pos"/"neg"], train/train.py:81
            got = M.CosineEmbeddingLoss(margin)(dev(q), other.to(DEV), torch.tensor(tgt).unsqueeze(0)).item()
            assert got == pytest.approx(want, rel=1e-5, abs=1e-7), (margin, tgt)
    m = M.validation_metrics(dev(q), dev(p), dev(n), dev(cls), margin=0.5)
    cos = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
    top1 = top3 = 0
    for idx in range(48):                                                # train/train.py:342-362, literally
        sim = cos(tq[idx].unsqueeze(0), tp)
        vals, inds = torch.topk(sim, k=3)
        top3 += int(cls[idx] in cls[inds.numpy()])
        top1 += int(cls[idx] == cls[inds[0].item()])
    assert m["top1"].item() == pytest.approx(top1 / 48) and m["top3"].item() == pytest.approx(top3 / 48)
    assert m["cos_sims"].item() == pytest.approx(cos(tq, tp).mean().item(), abs=1e-5)
    assert m["cos_unsims"].item() == pytest.approx(cos(tq, tn).mean().item(), abs=1e-5)


def test_score_booster_matches_reference_formulas():
    """utils/score_booster.py:1-37 over a whole tensor (reference: one python float at a time); fp32, same op order."""
    s = np.linspace(-0.2, 1.0, 1001, dtype=np.float32)
    t = torch.from_numpy(s).to(DEV)
    eps, alpha, thr = 0.3, 0.7, 0.55
    got = M.cos_sim_score_with_threshold(t, eps, alpha, thr).cpu().numpy()
    np.testing.assert_array_equal(got, orank.score_boost(s, eps, alpha, thr))
    np.testing.assert_array_equal(M.cos_sim_score_booster(t, eps, alpha, "for_pos").cpu().numpy(),
                                  orank.score_boost(s, eps, alpha, mode="for_pos"))
    np.testing.assert_array_equal(M.cos_sim_score_booster(t, eps, alpha, "for_neg").cpu().numpy(),
                                  orank.score_boost(s, eps, alpha, mode="for_neg"))
    # the published formulas in double on a few points (what the reference computes for python floats)
    for v in (0.1, 0.55, 0.9):
        want = (v + eps) / (eps + alpha) if v >= thr else abs((v + (alpha / eps)) / (2 * eps))
        g = float(M.cos_sim_score_with_threshold(torch.tensor([v], device=DEV), eps, alpha, thr)[0])
        assert abs(g - want) < 1e-6 * max(1.0, abs(want))
    with pytest.raises(M.MI355Error):
        M.cos_sim_score_booster(t, eps, alpha, "sideways")
    assert M.cos_sim_score_with_threshold(t[:0], eps, alpha, thr).shape == (0,)


def test_query_blocking_over_a_3m_row_gallery():
    """Q larger than the score-slab block (the slab S[qb][G] is capped at ~1 GiB, rank.hip query_block): 600 queries
    against 3M rows are processed as three blocks of 256; the result must not depend on the blocking (same tile shape
    for every block: compared against explicit 256-query calls) and match the oracle on probe rows."""
    Qn, Gn, d, k = 600, 3_000_000, 64, 5
    Q = M.synth_fill(Qn * d, 41, synth.NORMAL, DEV).view(Qn, d)
    G = M.l2_normalize_rows(M.synth_fill(Gn * d, 42, synth.NORMAL, DEV).view(Gn, d))
    v, i = M.cosine_topk(Q, G, k, gallery_is_normalized=True)
    for lo in range(0, Qn, 256):
        v2, i2 = M.cosine_topk(Q[lo:lo + 256], G, k, gallery_is_normalized=True)
        if v2.shape[0] > 128:            # same GEMM tile class as the blocked call (Q > 64 -> 128-query tiles)
            assert torch.equal(i[lo:lo + 256], i2) and torch.equal(v[lo:lo + 256], v2)
        else:
            assert torch.equal(i[lo:lo + 256], i2)
            torch.testing.assert_close(v[lo:lo + 256], v2, atol=SCORE_TOL, rtol=0)
    probes = [0, 255, 256, 511, 512, 599]
    qn = Q[probes].cpu().numpy().astype(np.float64)
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    g = G.cpu().numpy()
    for r, p in enumerate(probes):
        s = g @ qn[r].astype(np.float32)
        order = np.lexsort((np.arange(Gn), -s))[:k + 1]
        if np.min(s[order][:-1] - s[order][1:]) > 1e-5:
            assert i[p].cpu().tolist() == order[:k].tolist()
        np.testing.assert_allclose(v[p].cpu().numpy(), s[order[:k]], atol=SCORE_TOL)


@pytest.mark.parametrize("Qn,k", [(16, 3), (130, 3), (3, 3), (16, 150)])
def test_nan_scores_order_as_largest_like_torch(Qn, k):
    """torch.topk treats NaN as the largest value: a NaN embedding (bf16 overflow upstream) must come back with an
    in-range index, never as a pad entry that the class gathers would read out of bounds (fused-select, GEMV and
    slab + bitonic paths: Q > 4 / Q <= 4 / k > 8)."""
    Gn, d = 1000, 64
    Q = synth.normal(11, (Qn, d)).astype(np.float32)
    G = synth.normal(12, (Gn, d)).astype(np.float32)
    G[37, 5] = np.nan                      # one gallery row with a NaN -> its score is NaN for every query
    Q[1, :] = np.nan                       # one all-NaN query -> every score of that row is NaN
    v, i = M.cosine_topk(dev(Q), dev(G), k)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    assert i.min() >= 0 and i.max() < Gn, "a pad index leaked out"
    sim = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
    for q in range(Qn):
        want_v, want_i = torch.topk(sim(torch.from_numpy(Q[q][None]), torch.from_numpy(G)), k)
        if q == 1:
            assert np.isnan(v[q]).all()
            np.testing.assert_array_equal(i[q], np.arange(k))          # all tied (NaN): lower index first
            continue
        assert np.isnan(v[q, 0]) and i[q, 0] == 37 and bool(torch.isnan(want_v[0])) and int(want_i[0]) == 37
        np.testing.assert_allclose(v[q, 1:], want_v[1:].numpy(), atol=SCORE_TOL)
        np.testing.assert_array_equal(i[q, 1:], want_i[1:].numpy())
    # hit counting on these lists stays in bounds; an explicit pad index counts as a miss
    gcls = torch.arange(Gn) % 7
    counts = M.hit_counts(torch.from_numpy(i).to(DEV), torch.zeros(Qn, dtype=torch.int64), gcls)
    assert int(counts[1]) >= int(counts[0]) >= 0
    pad = torch.full((4, 3), np.iinfo(np.int64).max, dtype=torch.int64)
    assert M.hit_counts(pad.to(DEV), torch.zeros(4, dtype=torch.int64), gcls).tolist() == [0, 0]
    oc, oi, ov = M.distinct_class_topn(pad.to(DEV), torch.zeros(4, 3).to(DEV), gcls.to(DEV), 3)
    assert (oc.cpu() == -1).all()


def test_fused_select_matches_slab_path_bit_for_bit():
    """k <= 8 is selected inside the GEMM epilogue (no score slab); it must return exactly what selecting from the
    explicit score matrix returns (same tie rule), including ragged tiles and ties across column tiles."""
    Qn, Gn, d = 200, 5000, 96
    Q = synth.normal(21, (Qn, d)).astype(np.float32)
    G = synth.normal(22, (Gn, d)).astype(np.float32)
    G[130] = G[3]; G[4999] = G[3]; G[2000] = G[1999]          # exact duplicates in different column tiles
    for k in (1, 3, 8):
        v, i = M.cosine_topk(dev(Q), dev(G), k)
        S = M.cosine_scores(dev(Q), dev(G))
        v2, i2 = M.topk(S, k)
        assert torch.equal(v, v2) and torch.equal(i, i2), k


def _f64_cosine(q, g):
    qn = q.astype(np.float64) / np.maximum(np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True), 1e-6)
    gn = g.astype(np.float64) / np.maximum(np.linalg.norm(g.astype(np.float64), axis=1, keepdims=True), 1e-6)
    return qn @ gn.T


def test_split_bf16_gemm_is_fp32_equivalent(monkeypatch):
    """The default GEMM splits fp32 operands into three bf16 planes and keeps six of the nine products (rank.hip,
    split3): its scores must sit as close to the float64 cosine as the exact-fp32 MFMA chain's do (both are dominated by
    fp32 accumulation noise, ~1e-7), two orders below the 1e-5 of BASELINE.json, and the exact loop stays selectable."""
    Qn, Gn, d = 200, 6000, 1536
    rng = np.random.default_rng(7)
    q = synth.normal(41, (Qn, d)).astype(np.float32)
    g = (synth.normal(42, (Gn, d)) * rng.uniform(0.01, 30.0, (Gn, 1))).astype(np.float32)   # raw rows of very different norms
    g[17] = q[5] * 3.0                                           # a perfect match: score 1
    want = _f64_cosine(q, g)
    monkeypatch.delenv("MI355_RANK_EXACT_F32", raising=False)
    s_split = M.cosine_scores(dev(q), dev(g)).cpu().numpy().astype(np.float64)
    v_split, i_split = M.cosine_topk(dev(q), dev(g), 3)
    monkeypatch.setenv("MI355_RANK_EXACT_F32", "1")
    s_exact = M.cosine_scores(dev(q), dev(g)).cpu().numpy().astype(np.float64)
    v_exact, i_exact = M.cosine_topk(dev(q), dev(g), 3)
    monkeypatch.delenv("MI355_RANK_EXACT_F32")
    e_split, e_exact = np.abs(s_split - want).max(), np.abs(s_exact - want).max()
    assert e_exact < 5e-7 and e_split < 5e-7, (e_split, e_exact)
    assert e_split <= 2.0 * e_exact + 1e-7, (e_split, e_exact)
    assert np.abs(s_split - s_exact).max() < 1e-6          # two fp32 summation orders
    assert abs(s_split[5, 17] - 1.0) < 5e-7
    # the two loops rank identically wherever the float64 gaps exceed their noise
    srt = -np.sort(-want, axis=1)[:, :4]
    clear = (srt[:, :3] - srt[:, 1:4]).min(1) > 2e-6
    assert clear.sum() >= Qn - 5
    assert torch.equal(i_split[torch.from_numpy(clear)], i_exact[torch.from_numpy(clear)])
    np.testing.assert_array_equal(i_split.cpu().numpy()[clear], np.argsort(-want, axis=1)[clear][:, :3])


def test_split_scores_do_not_depend_on_tile_or_batch_shape():
    """A score is a function of its query row and gallery row only: the 64-row and 128-row tiles, the tail launch, a
    gallery slice and a query subset all give bit-identical values (what the sharded merge and bench.py's N > 1 self-check
    rely on)."""
    Qn, Gn, d = 256, 9000, 1536
    q, g = dev(synth.normal(51, (Qn, d))), dev(synth.normal(52, (Gn, d)))
    S = M.cosine_scores(q, g)
    assert torch.equal(M.cosine_scores(q[:64], g), S[:64])                  # 64-row tiles
    assert torch.equal(M.cosine_scores(q[100:105], g), S[100:105])          # five queries (padded tile)
    assert torch.equal(M.cosine_scores(q, g[4096:7000]), S[:, 4096:7000])   # a shard of the gallery
    v, i = M.cosine_topk(q, g, 3)
    v64, i64 = M.cosine_topk(q[:64], g, 3)
    assert torch.equal(v[:64], v64) and torch.equal(i[:64], i64)
    assert torch.equal(v, torch.gather(S, 1, i))


@pytest.mark.parametrize("Q,G,D,k", [(5, 257, 100, 3), (64, 1500, 1000, 3), (65, 129, 16, 1), (129, 4097, 36, 8), (200, 3000, 2560, 2)])
def test_split_loop_on_ragged_shapes_and_wide_dynamic_range(Q, G, D, k):
    """The split loop at shapes that end inside a tile / a k-step (D % 16 != 0, partial row and column tiles, 64- and
    128-row tiles) with rows whose norms span 19 orders of magnitude, all above the eps clamp (the bf16 planes keep fp32's exponent range):
    scores within 1e-5 of the float64 cosine, top-k identical to the float64 ranking wherever its gaps are clear."""
    rng = np.random.default_rng(Q * 1000 + D)
    q = (synth.normal(300 + Q, (Q, D)) * 10.0 ** rng.uniform(-4, 15, (Q, 1))).astype(np.float32)
    g = (synth.normal(400 + G, (G, D)) * 10.0 ** rng.uniform(-4, 15, (G, 1))).astype(np.float32)
    want = _f64_cosine(q, g)
    s = M.cosine_scores(dev(q), dev(g)).cpu().numpy().astype(np.float64)
    assert np.isfinite(s).all()
    np.testing.assert_allclose(s, want, rtol=0, atol=SCORE_TOL)
    assert np.abs(s - want).max() < 1e-6
    v, i = M.cosine_topk(dev(q), dev(g), k)
    order = np.argsort(-want, axis=1, kind="stable")[:, : k + 1]
    srt = np.take_along_axis(want, order, 1)
    clear = (srt[:, :-1] - srt[:, 1:]).min(1) > 2e-6
    assert clear.sum() >= Q * 0.9
    np.testing.assert_array_equal(i.cpu().numpy()[clear], order[clear][:, :k])
    np.testing.assert_allclose(v.cpu().numpy(), srt[:, :k], rtol=0, atol=SCORE_TOL)


def test_few_queries_with_rows_too_long_for_the_gemv():
    """Q <= 4 normally streams the gallery as a GEMV with the queries in LDS; rows longer than that LDS copy (60 KB) take
    the GEMM path, whose split planes must then be part of the workspace too."""
    Q, G, D = 3, 300, 6000
    q, g = synth.normal(61, (Q, D)), synth.normal(62, (G, D))
    want = _f64_cosine(q, g)
    s = M.cosine_scores(dev(q), dev(g)).cpu().numpy()
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-6)
    v, i = M.cosine_topk(dev(q), dev(g), 3)
    np.testing.assert_array_equal(i.cpu().numpy(), np.argsort(-want, axis=1)[:, :3])


@pytest.mark.parametrize("Q,G,D,k", [(256, 100000, 1536, 3), (64, 10000, 1536, 3), (16, 1000, 1536, 1), (300, 5000, 200, 8), (7, 333, 72, 2)])
def test_prepared_gallery_is_bit_identical_to_the_fp32_rows(Q, G, D, k):
    """mi355_gallery_prepare holds a resident gallery as the cosine GEMM's three bf16 planes (6 B per element, fragment order);
    mi355_rank_topk_prepared then multiplies them without any per-call work on the gallery side.  The planes are the same values
    the split loop derives in registers (split3 of the same normalised fp32 rows) and the six products are accumulated in
    the same order: values AND indices must be identical bits - at the metric's shape (two query blocks, whole rounds +
    64-row tail launch), for a single 64-query tile, ragged D / G / Q and the 3-stage MT = 1 ring."""
    import imageretrievalresearch_amd as M
    from imageretrievalresearch_amd import synth
    q = M.synth_fill(Q * D, 13, synth.NORMAL, DEV).view(Q, D)
    g = M.l2_normalize_rows(M.synth_fill(G * D, 5, synth.NORMAL, DEV).view(G, D))
    wv, wi = M.cosine_topk(q, g, k, gallery_is_normalized=True, idx_offset=11)
    p = M.PreparedGallery(g)
    v, i = p.search(q, k, idx_offset=11)
    assert torch.equal(v, wv) and torch.equal(i, wi)
    gal = M.Gallery(D, DEV)
    gal.add(g)                                  # (already unit rows: normalising again leaves them within 1 ulp, so compare to itself)
    a = gal.search(q, k)
    gal.prepare()
    b = gal.search(q, k)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_prepared_gallery_rejects_what_it_does_not_cover():
    import imageretrievalresearch_amd as M
    from imageretrievalresearch_amd import synth
    g = M.l2_normalize_rows(M.synth_fill(500 * 64, 5, synth.NORMAL, DEV).view(500, 64))
    p = M.PreparedGallery(g)
    q = M.synth_fill(8 * 64, 3, synth.NORMAL, DEV).view(8, 64)
    with pytest.raises(M.MI355Error):
        p.search(q, 150)                        # k > 8: the score-slab path needs the fp32 rows
    with pytest.raises(M.MI355Error):
        p.search(q[:2], 3)                      # Q <= 4: the GEMV streams the fp32 rows
