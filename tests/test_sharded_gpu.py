"""Row-sharded gallery with the REAL HIP backend: 2 and 3 ranks (gloo rendezvous, every rank on cuda:0 because
the box has one GPU; on a node the same code runs one rank per GPU over RCCL).  The 100k-row gallery of the
metric is split across ranks and the merged top-1/top-3 must equal the reference goldens bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import assert_topk_matches, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, key="g100k", check_full=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import imageretrievalresearch_amd as M
    from imageretrievalresearch_amd import synth
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = load_golden()
        qs, Qn, gs, Gn, d = (int(x) for x in G[f"{key}_meta"])
        lo, hi = Gn * rank // world, Gn * (rank + 1) // world
        shard = M.synth_fill((hi - lo) * d, gs, synth.NORMAL, "cuda:0", offset=lo * d).view(hi - lo, d)
        gal = M.ShardedGallery(shard)
        assert gal.total_rows == Gn and gal.offset == lo
        Ql = Qn // world                                   # each rank contributes its slice of the probe queries
        Qall = M.synth_fill(Qn * d, qs, synth.NORMAL, "cuda:0").view(Qn, d)
        v, i = gal.search(Qall[rank * Ql:(rank + 1) * Ql].contiguous(), 3)
        n = Ql * world
        ncert = assert_topk_matches(v.cpu().numpy(), i.cpu().numpy(), G[f"{key}_k3_val"][:n], G[f"{key}_k3_idx"][:n],
                                    G[f"{key}_k3_gap"][:n], float(G["cert_gap"]), f"sharded {key} world={world}")
        if not check_full:
            out[rank] = bool(ncert >= n - 2)
            return
        # and bit-identical to the unsharded HIP result (same kernels, same tie rule)
        full = M.synth_fill(Gn * d, gs, synth.NORMAL, "cuda:0").view(Gn, d)
        fv, fi = M.cosine_topk(Qall[:n], M.l2_normalize_rows(full), 3, gallery_is_normalized=True)
        out[rank] = bool(torch.equal(fi, i) and torch.equal(fv, v) and ncert >= n - 2)
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_100k_gallery_matches_goldens_and_single_gpu(world):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_sharded_1m_gallery_matches_goldens():
    """BASELINE configs[4] gallery size (1M rows) row-sharded over 2 ranks: merged top-3 equals the reference goldens."""
    if "g1m_meta" not in load_golden().files:
        pytest.skip("1M-row goldens not generated")
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, "g1m", False), nprocs=2, join=True)
    assert all(out.get(r) for r in range(2)), dict(out)


def test_bench_n2_branch_runs_and_checks_itself():
    """The N > 1 branch of bench.py (all-gather of queries, packed candidate all-gather, library-side merge, the 64-query
    self-check against the unsharded gallery, MAX-over-ranks timing) in fresh child processes: two gloo ranks on this
    one GPU (RCCL refuses two ranks on one device; the driver's real N > 1 run is the first to use it).  rc must be 0 and
    rank 0 must print one JSON line with the contract fields for n_gpus = 2."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "roofline" in d and "cpu_baseline" not in d          # the host baseline is an N = 1 field
