"""RexNet (rexnet_150 = the reference's default model, rexnet_200 = BASELINE configs[2]): HIP executor vs the CPU oracle.
Odd channel counts (54, 77, 167, ...) exercise the pad-to-8 layout and the partial-channel shortcut.
Backbone parity is UNPINNED against timm itself (see oracle/__init__.py)."""
import numpy as np
import pytest
import torch

import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import rank as orank, rexnet
from test_effnet_gpu import images, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_TAP_SIM = 2.5e-2      # ReLU6 clipping + 16 blocks: bf16 rounding flips compound a little faster than in effnet
TOL_EMB_FP32 = 5e-2


@pytest.mark.parametrize("name,wm", [("rexnet_150", 1.5), ("rexnet_200", 2.0)])
def test_rexnet_blocks_and_embedding_match_oracle(name, wm):
    sd = rexnet.init_state_dict(4, wm)
    model = M.create_model(name).to(DEV).eval()
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(images(21, 2))
    taps = {}
    want = rexnet.forward_features(sd, x, wm, sim_bf16=True, taps=taps)
    taps["head"] = want
    model.enable_taps(True)
    got = model.forward_features(x.to(DEV))
    worst = 0.0
    for tname, ref in taps.items():
        t = model.read_tap(tname).cpu()
        assert t.shape == ref.shape, (tname, t.shape, ref.shape)
        e = rel(t, ref)
        worst = max(worst, e)
        assert e < TOL_TAP_SIM, f"{name} tap {tname}: rel L2 {e:.3e}"
    model.enable_taps(False)
    print(name, "worst tap rel L2", worst)
    assert got.shape == want.shape and rel(got.cpu(), want) < TOL_TAP_SIM
    # train/train.py:194-195 call shape: fm = forward_features(x); lbl = head(fm)
    lbl = model.head(got)                          # ClassifierHead -> mi355_pool_linear (HIP), not torch eager
    assert lbl.shape == (2, 1000)
    want_logits = rexnet.forward(sd, x, wm, sim_bf16=True)
    out = model(x.to(DEV))
    assert rel(out.cpu(), want_logits) < 3e-2
    assert rel(lbl.cpu(), want_logits) < 3e-2      # forward_features -> head(fm) == model(x), both against the oracle
    assert rel(lbl.cpu(), out.cpu()) < 2e-3        # same rounding points, different summation order
    pooled_only = M.models.pool_linear(got)        # get_fm (train/train.py:84-103)
    assert torch.equal(pooled_only, model.embed(x.to(DEV))[0])
    f32 = rexnet.forward_features(sd, x, wm).mean((2, 3))
    pooled, _ = model.embed(x.to(DEV))
    assert rel(pooled.cpu(), f32) < TOL_EMB_FP32
    assert torch.nn.functional.cosine_similarity(pooled.cpu(), f32).min() > 0.998
    assert model.head.fc.in_features == model.num_features         # inference/inference.py:140


def test_rexnet200_contrastive_and_pair_cosine_on_embeddings():
    """BASELINE configs[2]: contrastive_loss.py / cosine parity on rexnet_200 (B, 2560) embeddings."""
    sd = rexnet.init_state_dict(4, 2.0, num_classes=0)
    model = M.create_model("rexnet_200", num_classes=0).to(DEV).eval()
    model.load_state_dict(sd, strict=True)
    xa = torch.from_numpy(images(31, 4)).to(DEV)
    xb = torch.from_numpy(images(32, 4)).to(DEV)
    ea, eb = model(xa), model(xb)
    assert ea.shape == (4, 2560)
    na, nb = ea.cpu().numpy(), eb.cpu().numpy()
    for label, mean in ((1.0, True), (0.0, False)):
        got = M.ContrastiveLoss(0.5)(ea, eb, label, mean).item()
        want = float(orank.contrastive_loss(na, nb, label, 0.5, mean))
        assert got == pytest.approx(want, rel=1e-5, abs=1e-7)
    np.testing.assert_allclose(M.pair_cosine(ea, eb).cpu().numpy(), orank.pair_cosine(na, nb), atol=1e-5)
    v, i = M.cosine_topk(ea, eb, 3)
    wv, wi = orank.rank_topk(na, nb, 3)
    np.testing.assert_allclose(v.cpu().numpy(), wv, atol=1e-5)


def test_rexnet_head_identity_and_determinism():
    model = M.create_model("rexnet_150").to(DEV).eval()
    x = torch.from_numpy(images(41, 3)).to(DEV)
    a = model(x)
    assert torch.equal(a, model(x)) and a.shape == (3, 1000)
    model.head = torch.nn.Identity()                    # notebook raw :190 idiom
    e = model(x)
    assert e.shape == (3, 1920)
    model.set_option("fuse", 0)
    e2 = model(x)
    model.set_option("fuse", 1)
    assert rel(e2, e) < 1e-2                            # fused and unfused executors agree to rounding


def test_rexnet200_full_batch_properties_b256():
    """BASELINE configs[2] size (rexnet_200, bs = 256): determinism, bit-exact permutation equivariance, agreement
    with the same images in a batch of 4, and the cosine path on those embeddings (rank of the batch against itself:
    every image's nearest neighbour is itself with score 1)."""
    model = M.create_model("rexnet_200", num_classes=0, seed=4).to(DEV).eval()
    B = 256
    x = M.synth_fill(B * 3 * 224 * 224, 78, synth.UNIFORM, DEV).view(B, 3, 224, 224)
    a = model(x)
    assert a.shape == (B, 2560) and torch.isfinite(a).all()
    assert torch.equal(a, model(x))
    perm = torch.from_numpy(np.random.RandomState(5).permutation(B)).to(DEV)
    assert torch.equal(model(x[perm].contiguous()), a[perm])
    assert rel(model(x[:4].contiguous()).cpu(), a[:4].cpu()) < 5e-3
    v, i = M.cosine_topk(a, a, 1)
    np.testing.assert_allclose(v.cpu().numpy(), 1.0, atol=1e-5)
    # uniform-noise images through a random-init net give near-identical embeddings: an index may only differ from the
    # diagonal where another image's score is within the score tolerance of 1
    s = M.cosine_scores(a, a)
    off = (i[:, 0] != torch.arange(B, device=DEV)).nonzero().flatten()
    for r in off.tolist():
        assert float(s[r, int(i[r, 0])]) > 1.0 - 1e-5


TOL_BLOCK_ISOLATED = 2e-3     # ONE block fed the oracle's own input: a fraction of a bf16 ulp (2^-8 = 3.9e-3) of relative L2


@pytest.mark.parametrize("name,wm,B", [("rexnet_200", 2.0, 2), ("rexnet_200", 2.0, 256), ("rexnet_150", 1.5, 256)])
def test_each_rexnet_block_on_the_oracles_own_input(name, wm, B):
    """BASELINE configs[2] at the M where its kernels are chosen: every layer group between two taps (stem -> each of the 16
    LinearBottlenecks -> head) runs ALONE on the oracle's bf16-rounded activation of the previous tap
    (mi355_model_run_between_taps), so an error cannot hide behind the compounding tolerance of the whole-network test.
    B = 256 repeats the two oracle images 128 times: 256 * h * w rows is where the executor picks the three-workgroup
    short-K GEMM, the gated DMA GEMM, the band / late fused kernels and (from round 3) the whole-block kernel for the
    padded channel counts; every repeat must be bit-identical to the first."""
    sd = rexnet.init_state_dict(4, wm)
    model = M.create_model(name).to(DEV).eval()
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(images(23, 2))
    taps = {}
    want_head = rexnet.forward_features(sd, x, wm, sim_bf16=True, taps=taps)
    taps["head"] = want_head
    order = list(taps.keys())
    assert order[0] == "stem" and order[-1] == "head" and len(order) == 18
    model.enable_taps(True)
    worst = ("", 0.0)
    for prev, cur in zip(order[:-1], order[1:]):
        src = taps[prev].to(DEV)
        if B > 2:
            src = src.repeat(B // 2, 1, 1, 1).contiguous()
        model.run_between_taps(prev, cur, src)
        got = model.read_tap(cur)
        del src
        if B > 2:
            g = got.view(B // 2, 2, *got.shape[1:])
            assert torch.equal(g[0], g[1]) and torch.equal(g[0], g[-1]), f"{cur}: result depends on the batch position"
            got = g[0]
        e = rel(got.cpu(), taps[cur])
        worst = max(worst, (cur, e), key=lambda p: p[1])
        assert e < TOL_BLOCK_ISOLATED, f"{name} {prev} -> {cur} at B={B}: rel L2 {e:.3e}"
        del got
    model.enable_taps(False)
    print(f"{name} B={B}: worst isolated block {worst}")


@pytest.mark.parametrize("name,B,mb", [("rexnet_150", 256, 100), ("rexnet_200", 120, 70)])
def test_rexnet_chunking_does_not_change_the_bits(name, B, mb):
    """Same rule as tests/test_effnet_gpu.py::test_chunking_does_not_change_the_bits, on RexNet's own kernel choices: the whole-block
    kernel (from 96 images of the CALLER's batch), the row-sweep classes (shape only), the wave-private projection kernel (from 2^19
    rows of the chunk - same k order as the tiled kernel, so the same bits) and the split-K projections.  Microbatches that straddle
    the thresholds and two concurrent lanes must reproduce the unchunked forward exactly."""
    from imageretrievalresearch_amd import synth
    model = M.create_model(name, num_classes=0, seed=4).to(DEV).eval()
    x = M.synth_fill(B * 3 * 224 * 224, 33, synth.UNIFORM, DEV).view(B, 3, 224, 224)
    want = model(x).clone()
    try:
        model.set_option("microbatch", mb)
        assert torch.equal(model(x), want)
        model.set_option("microbatch", 0)
        model.set_option("lanes", 2)
        assert torch.equal(model(x), want)
    finally:
        model.set_option("microbatch", 0)
        model.set_option("lanes", 1)

