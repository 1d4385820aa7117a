"""EfficientNet-B3a: HIP executor (through the C ABI / timm-shaped Module) against the CPU oracle.

Two references (oracle/effnet.py): ``sim_bf16=True`` reproduces the kernels' bf16 rounding points, so
every layer must agree to a few bf16 ulps; ``sim_bf16=False`` is the reference's fp32 CPU semantics
(autocast is a no-op on CPU), against which bf16 storage costs ~0.5 % relative error on the embedding.
Backbone parity is UNPINNED against timm itself (not installed, no reference fixtures) — see oracle/__init__.py.
"""
import numpy as np
import pytest
import torch

import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# tolerances (relative L2 over a whole tensor): per-layer vs the bf16-simulating oracle, and the
# final embedding vs the fp32 oracle
TOL_LAYER_SIM = 1.5e-2   # whole-network taps: rounding flips compound with depth (1 bf16 ulp = 3.9e-3)
TOL_EMB_SIM = 1e-2
TOL_EMB_FP32 = 2e-2


def images(seed, B, H=224, W=224):
    """Uniform noise plus a per-image low-frequency pattern so embeddings differ between images."""
    x = synth.uniform(seed, (B, 3, H, W))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for b in range(B):
        x[b] = 0.5 * x[b] + 0.5 * (0.5 + 0.5 * np.sin(xx / (3.0 + b) + b) * np.cos(yy / (5.0 + 2 * b)))[None]
    return x.astype(np.float32)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def setup():
    sd = effnet.init_state_dict(2)
    model = M.create_model("efficientnet_b3a").to(DEV).eval()
    model.load_state_dict(sd, strict=True)
    return sd, model


def test_every_block_matches_bf16_sim_oracle(setup):
    sd, model = setup
    x = torch.from_numpy(images(1, 2))
    taps = {}
    want = effnet.forward_features(sd, x, sim_bf16=True, taps=taps)
    taps["head"] = want
    model.enable_taps(True)
    got = model.forward_features(x.to(DEV))
    worst = ("", 0.0)
    for name, ref in taps.items():
        t = model.read_tap(name).cpu()
        assert t.shape == ref.shape, (name, t.shape, ref.shape)
        e = rel(t, ref)
        worst = max(worst, (name, e), key=lambda p: p[1])
        print(f"tap {name:12s} rel L2 vs bf16-sim oracle {e:.3e}")
        assert e < TOL_LAYER_SIM, f"tap {name}: rel L2 {e:.3e}"
    model.enable_taps(False)
    assert got.shape == (2, 1536, 7, 7)
    assert rel(got.cpu(), want) < TOL_LAYER_SIM
    print("worst tap", worst)


def test_other_input_size_takes_the_general_kernels(setup):
    """200 x 200 images: the stem output is 100 x 100 (10 000 pixels = 312.5 tiles of 32: the MFMA depthwise kernel's partial
    last tile), and none of the row-sweep kernel's shape classes (112 / 56 / 28) applies, so the band / unfused kernels run -
    same tap tolerances against the oracle."""
    sd, model = setup
    x = torch.from_numpy(images(9, 2, H=200, W=200))
    taps = {}
    want = effnet.forward_features(sd, x, sim_bf16=True, taps=taps)
    model.enable_taps(True)
    got = model.forward_features(x.to(DEV))
    for name in ("stem", "blocks.0.0", "blocks.0.1", "blocks.1.0", "blocks.2.0", "blocks.3.1", "blocks.6.1"):
        e = rel(model.read_tap(name).cpu(), taps[name])
        assert e < TOL_LAYER_SIM, f"tap {name} at 200x200: rel L2 {e:.3e}"
    model.enable_taps(False)
    assert got.shape == want.shape and rel(got.cpu(), want) < TOL_LAYER_SIM


def test_embedding_vs_fp32_reference_semantics(setup):
    sd, model = setup
    x = torch.from_numpy(images(3, 4))
    f32 = effnet.pool(effnet.forward_features(sd, x, sim_bf16=False))
    sim = effnet.pool(effnet.forward_features(sd, x, sim_bf16=True))
    pooled, logits = model.embed(x.to(DEV))
    assert rel(pooled.cpu(), sim) < TOL_EMB_SIM
    assert rel(pooled.cpu(), f32) < TOL_EMB_FP32
    cos = torch.nn.functional.cosine_similarity(pooled.cpu(), f32)
    assert cos.min() > 0.9995
    # logits = classifier(pooled): forward() and embed() agree, and match the oracle's forward
    want_logits = effnet.forward(sd, x, sim_bf16=True)
    out = model(x.to(DEV))
    assert torch.equal(out, logits)
    assert rel(out.cpu(), want_logits) < 1e-2
    # forward_features + get_fm (train/train.py:101-103) equals the pooled output
    fm = model.forward_features(x.to(DEV))
    gp = torch.reshape(torch.nn.AvgPool2d((fm.shape[2], fm.shape[3]))(fm), (-1, fm.shape[1]))
    torch.testing.assert_close(gp, pooled, rtol=1e-5, atol=1e-6)


def test_deterministic_and_batch_invariant(setup):
    _, model = setup
    x = torch.from_numpy(images(5, 6)).to(DEV)
    a = model(x)
    b = model(x)
    assert torch.equal(a, b)
    c = model(x[2:5])
    assert torch.equal(a[2:5], c)           # each image's result does not depend on its batch neighbours
    model.set_option("microbatch", 4)
    d = model(x)
    model.set_option("microbatch", 0)
    assert torch.equal(a, d)


@pytest.mark.parametrize("B,mb", [(256, 100), (120, 70), (90, 30)])
def test_chunking_does_not_change_the_bits(setup, B, mb):
    """the whole-block kernel (chosen from 96 images up) and the split-K projections (chosen up to 16 384 rows, i.e.
    83 images at 14x14) round differently from their alternatives, and both used to be chosen by the CHUNK a forward is cut
    into - a microbatch remainder or a lane embedded differently from the same image in the unchunked batch.  The choice now
    follows the caller's whole batch: microbatches that straddle either threshold, and two concurrent lanes, give the same bits."""
    _, model = setup
    x = M.synth_fill(B * 3 * 224 * 224, 31, synth.UNIFORM, DEV).view(B, 3, 224, 224)
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


def test_full_batch_properties_b256(setup):
    """BASELINE configs[1] size (bs = 256), through size-independent properties: run-to-run determinism, permutation
    equivariance (bit-exact: no result depends on the image's position or neighbours), and agreement with the same
    images embedded in a batch of 4 (different kernels are selected at small M, so only to rounding noise)."""
    _, model = setup
    B = 256
    x = M.synth_fill(B * 3 * 224 * 224, 77, synth.UNIFORM, DEV).view(B, 3, 224, 224)
    a = model(x)
    assert a.shape[0] == B and torch.isfinite(a).all()
    assert torch.equal(a, model(x))
    perm = torch.from_numpy(np.random.RandomState(3).permutation(B)).to(DEV)
    assert torch.equal(model(x[perm].contiguous()), a[perm])
    small = model(x[:4].contiguous())
    assert rel(small.cpu(), a[:4].cpu()) < 5e-3


def test_num_classes_zero_and_head_swaps(setup):
    sd, _ = setup
    x = torch.from_numpy(images(7, 2)).to(DEV)
    m0 = M.create_model("efficientnet_b3a", num_classes=0).to(DEV).eval()
    missing = m0.load_state_dict(sd, strict=False)     # classifier.* unexpected, nothing missing
    assert not missing.missing_keys and set(missing.unexpected_keys) == {"classifier.weight", "classifier.bias"}
    emb = m0(x)
    assert emb.shape == (2, 1536)
    m1 = M.create_model("efficientnet_b3a").to(DEV).eval()
    m1.load_state_dict(sd)
    p, _ = m1.embed(x)
    assert torch.equal(emb, p)
    m1.classifier = torch.nn.Identity()                # notebook raw :190 / train_vit_triplet.py:357 idiom
    assert torch.equal(m1(x), emb)
    lin = torch.nn.Linear(1536, 125).to(DEV)           # inference/inference.py:141 idiom
    m1.classifier = lin
    out = m1(x)
    assert out.shape == (2, 125)
    want = torch.nn.functional.linear(emb.bfloat16().float(), lin.weight.bfloat16().float(), lin.bias)
    torch.testing.assert_close(out, want, rtol=2e-3, atol=2e-3)


def test_conv_input_wrapper_matches_torch(setup):
    sd, model = setup
    x = torch.from_numpy(images(9, 2))
    wrapped = M.models.with_conv_input(model)
    keys = list(wrapped.state_dict().keys())
    assert keys[0] == "0.0.weight" and keys[1] == "1.conv_stem.weight"     # inference.py:103-105 key surface
    w = wrapped[0][0].weight.detach().cpu()
    pre = torch.nn.functional.silu(torch.nn.functional.conv2d(x, w, padding=1))
    got_pre = wrapped[0](x.to(DEV)).cpu()
    torch.testing.assert_close(got_pre, pre, rtol=1e-5, atol=1e-5)
    want = effnet.forward(sd, pre, sim_bf16=True)
    got = wrapped(x.to(DEV)).cpu()
    assert rel(got, want) < 1e-2


def test_odd_input_sizes_and_errors(setup):
    sd, model = setup
    x = torch.from_numpy(images(11, 1, 160, 192))
    want = effnet.forward_features(sd, x, sim_bf16=True)
    got = model.forward_features(x.to(DEV))
    assert got.shape == want.shape == (1, 1536, 5, 6)
    assert rel(got.cpu(), want) < TOL_LAYER_SIM
    with pytest.raises(M.MI355Error):
        model(torch.zeros(1, 3, 224, 224))             # CPU tensor: no fallback
    with pytest.raises(M.MI355Error):
        model(torch.zeros(1, 1, 224, 224, device=DEV))
    with pytest.raises(M.MI355Error):
        M.create_model("efficientnet_b3a", pretrained=True)


def test_reload_in_eval_mode_uses_the_new_weights(setup):
    """forward, then load new weights through the PARENT wrapper in eval mode, forward again: the
    second output must be the oracle's for the new weights (a stale pack would reproduce the first)."""
    sd, _ = setup
    x = torch.from_numpy(images(13, 2))
    model = M.create_model("efficientnet_b3a", num_classes=0)
    wrapped = M.models.with_conv_input(model).to(DEV).eval()
    w0 = torch.zeros(3, 3, 3, 3)
    for c in range(3):
        w0[c, c, 1, 1] = 1.5                               # conv_input ~ identity x 1.5 before its SiLU
    full = {"0.0.weight": w0}
    full.update({"1." + k: v for k, v in sd.items() if not k.startswith("classifier.")})
    wrapped.load_state_dict(full)
    first = wrapped(x.to(DEV)).cpu()
    sd2 = effnet.init_state_dict(3)
    full2 = {"0.0.weight": w0}
    full2.update({"1." + k: v for k, v in sd2.items() if not k.startswith("classifier.")})
    wrapped.load_state_dict(full2)                         # parent load, eval mode, after a forward
    second = wrapped(x.to(DEV)).cpu()
    pre = torch.nn.functional.silu(torch.nn.functional.conv2d(x, w0, padding=1))
    want2 = effnet.pool(effnet.forward_features(sd2, pre, sim_bf16=True))
    assert rel(second, want2) < TOL_EMB_SIM
    assert rel(first, want2) > 0.1                         # the two weight sets really differ
    with torch.no_grad():                                  # in-place write in eval mode is picked up too
        wrapped[1].conv_stem.weight.zero_()
    third = wrapped(x.to(DEV)).cpu()
    # a zero stem makes the network blind to its input: both images now embed identically (and not as before)
    assert torch.equal(third[0], third[1]) and not torch.equal(second[0], second[1]) and not torch.equal(third, second)


def test_lightning_checkpoint_to_gpu_forward(setup, tmp_path):
    """f-2 on the GPU (inference/inference.py:98,113-124): a Lightning-style .ckpt with ``model.1.`` / ``model.0.0``
    keys -> load_checkpoint(conv_input=True) -> .to(device) -> forward, against the oracle on the conv_input output."""
    sd, _ = setup
    w0 = synth.normal(5, (3, 3, 3, 3)).astype(np.float32) * 0.3
    ck = {"model.1." + k: v for k, v in sd.items()}
    ck["model.0.0.weight"] = torch.from_numpy(w0)
    path = tmp_path / "epoch=3-val_loss=0.10-cos_sims=0.90-val_top1=0.80.ckpt"
    torch.save({"state_dict": ck, "epoch": 3}, path)
    model = M.load_checkpoint(str(path), "efficientnet_b3a", conv_input=True)
    assert not model.load_report.missing_keys and not model.load_report.unexpected_keys
    model = model.to(DEV).eval()
    x = torch.from_numpy(images(15, 2))
    got = model(x.to(DEV)).cpu()                           # default 1000-way head kept, as the reference does (:102)
    pre = torch.nn.functional.silu(torch.nn.functional.conv2d(x, torch.from_numpy(w0), padding=1))
    want = effnet.forward(sd, pre, sim_bf16=True)
    assert got.shape == want.shape == (2, 1000)
    assert rel(got, want) < 1e-2


def test_block_kernel_agrees_with_the_unfused_chain(setup):
    """The whole-block kernel (14x14 / 7x7 stages) against expand -> depthwise -> SE -> gated projection as separate
    kernels on the same weights: same rounding points, different summation orders - the first fused block may differ
    from the chain only by a few bf16 roundings, and nothing may depend on the position in the batch."""
    sd, model = setup
    x = torch.from_numpy(images(17, 3)).to(DEV)
    names = ["blocks.3.0", "blocks.3.1", "blocks.4.0", "blocks.5.0", "blocks.6.1", "head"]
    taps = {}
    model.set_option("fuse_block_min_batch", 1)      # the default only picks the block kernel for >= 96 images
    for opt in (0, 1):
        model.set_option("fuse_block", opt)
        model.enable_taps(True)
        model.forward_features(x)
        taps[opt] = {n: model.read_tap(n).cpu() for n in names}
        model.enable_taps(False)
    model.set_option("fuse_block", 1)
    assert torch.equal(taps[0]["blocks.3.0"], taps[1]["blocks.3.0"])      # last block before the fused stages
    first = rel(taps[1]["blocks.3.1"], taps[0]["blocks.3.1"])
    assert first < 2.5e-3, first                                           # < 1 bf16 ulp relative L2
    for n in names[2:]:
        assert rel(taps[1][n], taps[0][n]) < 1.2e-2, n
    xx = x[:1].repeat(5, 1, 1, 1).contiguous()                            # same image at five batch positions
    out = model.forward_features(xx)
    model.set_option("fuse_block_min_batch", 96)
    for i in range(1, 5):
        assert torch.equal(out[0], out[i]), i


def test_sweep_kernel_agrees_with_the_unfused_pair(setup):
    """The row-sweep kernel (expand + MFMA depthwise of the 112x112 .. 28x28 blocks) against the expand GEMM and the
    depthwise kernel run separately on the same weights: both round the expanded tensor to bf16 and accumulate the taps
    in fp32, so the first swept block may differ by a few bf16 roundings only; nothing may depend on the batch position
    (the kernel walks its channel slabs from an image-dependent start)."""
    sd, model = setup
    x = torch.from_numpy(images(23, 3)).to(DEV)
    names = ["blocks.0.1", "blocks.1.0", "blocks.1.1", "blocks.2.0", "blocks.2.1", "blocks.3.0"]
    taps = {}
    for opt in (0, 1):
        model.set_option("fuse_sweep", opt)
        model.set_option("fuse_band", 0 if opt == 0 else 2)
        model.enable_taps(True)
        model.forward_features(x)
        taps[opt] = {n: model.read_tap(n).cpu() for n in names}
        model.enable_taps(False)
    model.set_option("fuse_sweep", 1)
    model.set_option("fuse_band", 2)
    assert torch.equal(taps[0]["blocks.0.1"], taps[1]["blocks.0.1"])      # last block before the swept stages
    first = rel(taps[1]["blocks.1.0"], taps[0]["blocks.1.0"])
    assert first < 1e-3, first
    for n in names[2:]:
        assert rel(taps[1][n], taps[0][n]) < 6e-3, n
    xx = x[:1].repeat(11, 1, 1, 1).contiguous()                           # same image at eleven batch positions (> 8 XCDs)
    out = model.forward_features(xx)
    for i in range(1, 11):
        assert torch.equal(out[0], out[i]), i


def test_projection_kernel_choice_does_not_show_in_the_result(setup):
    """The narrow SE-gated projections run in the wave-private streaming kernel (k_proj_lds) once M = B*h*w >= 2^19 rows
    and in the tiled kernel below that; both accumulate the k-steps in ascending order from zero and add the bias last,
    so the block output for an image must be BIT-identical whichever kernel its batch size selects (block 1.1: 56x56,
    192 -> 32 gated + residual; block 0.1: 112x112, 24 -> 24 gated + residual)."""
    _, model = setup
    model.enable_taps(True)
    for prev, cur, c, hw in (("blocks.1.0", "blocks.1.1", 32, 56), ("blocks.0.0", "blocks.0.1", 24, 112)):
        big = torch.from_numpy(synth.normal(31, (8, c, hw, hw)).astype(np.float32)).to(DEV).bfloat16().float()
        model.run_between_taps(prev, cur, big)                               # M = 8*hw*hw < 2^19: tiled kernel
        want = model.read_tap(cur).clone()
        rep = big.repeat(32, 1, 1, 1).contiguous()                           # 256 images: streaming kernel
        model.run_between_taps(prev, cur, rep)
        got = model.read_tap(cur)
        assert torch.equal(got[:8], want) and torch.equal(got[-8:], want), cur
        del rep, got
    model.enable_taps(False)


TOL_BLOCK_ISOLATED = 1.5e-3   # measured <= 6e-4: a fraction of one bf16 ulp (2^-8 = 3.9e-3) of relative L2 for ONE block fed the oracle's own input


@pytest.mark.parametrize("B", [2, 256])
def test_each_block_on_the_oracles_own_input(setup, B):
    """Every layer group between two taps (stem -> block -> ... -> head) is run ALONE on the oracle's bf16-rounded
    activation of the previous tap (mi355_model_run_between_taps), so an error cannot hide behind the compounding
    tolerance of the whole-network test.  B = 256 repeats the two oracle images 128 times: that is the M (= 256*h*w
    rows) at which the executor picks its B=256 kernels (DMA GEMM, streaming GEMM, band / whole-block fusion), and
    every repeat must be bit-identical to the first."""
    sd, model = setup
    x = torch.from_numpy(images(19, 2))
    taps = {}
    want_head = effnet.forward_features(sd, x, sim_bf16=True, taps=taps)
    taps["head"] = want_head
    order = list(taps.keys())
    assert order[0] == "stem" and order[-1] == "head" and len(order) == 28
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
        assert e < TOL_BLOCK_ISOLATED, f"{prev} -> {cur} at B={B}: rel L2 {e:.3e}"
    model.enable_taps(False)
    print(f"B={B}: worst isolated block {worst}")


def test_roctx_ranges_do_not_change_results(setup):
    """set_option("roctx", 1) wraps every executor launch and rank phase in a roctxRangePush/Pop pair (marker library
    looked up at run time, no-ops without it): results must be unaffected and the switch must turn off again."""
    sd, model = setup
    x = torch.from_numpy(images(91, 4)).to(DEV)
    want = model(x)
    wv, wi = M.cosine_topk(want, want, 2)
    model.set_option("roctx", 1)
    try:
        got = model(x)
        v, i = M.cosine_topk(got, got, 2)
    finally:
        model.set_option("roctx", 0)
    assert torch.equal(got, want) and torch.equal(v, wv) and torch.equal(i, wi)


@pytest.mark.parametrize("name", ["efficientnet_b3a", "rexnet_150", "rexnet_200"])
def test_head_conv_with_the_pool_in_its_epilogue_is_bit_identical(name):
    """a6 (get_fm, train/train.py:84-103): when the caller wants the pooled embedding, conv_head / features.16 + bias + SiLU + the
    global average pool run as ONE kernel per (image, 128 channels) - a tile never spans two images, and the sum over the 49
    pixels keeps k_gap's sequential order - so embeddings AND logits must be the very bits of the conv -> k_gap path."""
    model = M.create_model(name, num_classes=1000, seed=9).to(DEV).eval()
    x = torch.from_numpy(images(43, 5)).to(DEV)
    model.set_option("fuse_head_gap", 0)
    emb0, log0 = model.embed(x)
    out0 = model(x)
    model.set_option("fuse_head_gap", 1)
    emb1, log1 = model.embed(x)
    out1 = model(x)
    assert torch.equal(emb0, emb1) and torch.equal(log0, log1) and torch.equal(out0, out1)
    # the un-pooled map is still what forward_features returns, and pooling it by hand gives the same embedding
    fm = model.forward_features(x)
    assert torch.equal(M.models.pool_linear(fm), emb1)
