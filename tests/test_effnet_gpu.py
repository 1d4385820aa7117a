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
