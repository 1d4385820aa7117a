"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo processes drive ShardedGallery's collective plumbing
(query all-gather, global index offsets, candidate all-gather layout, merge order).  The compute backend
is injected from the oracle here — on GPUs the default backend is the HIP library (tests/test_rank_gpu.py
checks those kernels and merge_topk; this file checks that sharded == unsharded, bit for bit)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from imageretrievalresearch_amd import synth
from imageretrievalresearch_amd.sharded import ShardedGallery
from oracle import rank as orank


class OracleOps:
    @staticmethod
    def normalize(rows):
        return torch.from_numpy(orank.l2_normalize_rows(rows.numpy()))

    @staticmethod
    def local_topk(queries, gallery_normalized, k, idx_offset):
        S = orank.cosine_scores(queries.numpy(), gallery_normalized.numpy())
        v, i = orank.topk_rows(S, k)
        return torch.from_numpy(v), torch.from_numpy(i + idx_offset)

    @staticmethod
    def pack(vals, idx, Q, k, device):
        packed = np.empty((Q, k, 2), dtype=np.int32)
        packed[:, :, 0] = np.float32(-np.inf).view(np.int32)
        packed[:, :, 1] = -1
        if vals is not None:
            kk = vals.shape[1]
            packed[:, :kk, 0] = vals.numpy().astype(np.float32).view(np.int32)
            packed[:, :kk, 1] = idx.numpy().astype(np.int32)
        return torch.from_numpy(packed)

    @staticmethod
    def merge_packed(packed, shard_offsets, k):
        p = packed.numpy()                                    # (world, Q, k, 2)
        world, Q = p.shape[0], p.shape[1]
        v = np.ascontiguousarray(p[..., 0]).view(np.float32).transpose(1, 0, 2).reshape(Q, world * k)
        li = p[..., 1].astype(np.int64)
        gi = np.where(li >= 0, li + shard_offsets.numpy().reshape(world, 1, 1), np.int64(2) ** 62)
        i = gi.transpose(1, 0, 2).reshape(Q, world * k)
        order = np.lexsort((i, -v), axis=1)[:, :k]          # score desc, then lower global index
        return (torch.from_numpy(np.take_along_axis(v, order, 1)), torch.from_numpy(np.take_along_axis(i, order, 1)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bounds, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, Ql = 64, 5
        G = synth.normal(71, (bounds[-1], D))
        G[40] = G[7]                                         # exact cross-shard tie: lower global index must win
        G[bounds[1] + 3] = G[7] if world > 1 else G[bounds[1] - 1]
        Qall = synth.normal(72, (world * Ql, D))
        Qall[0] = G[7]
        shard = torch.from_numpy(G[bounds[rank]:bounds[rank + 1]].copy())
        gal = ShardedGallery(shard, ops=OracleOps)
        assert gal.total_rows == bounds[-1] and gal.offset == bounds[rank]
        v, i = gal.search(torch.from_numpy(Qall[rank * Ql:(rank + 1) * Ql].copy()), k)
        # unsharded reference through the SAME backend (gallery normalised once, then scored)
        wv, wi = OracleOps.local_topk(torch.from_numpy(Qall), OracleOps.normalize(torch.from_numpy(G)), k, 0)
        ok = bool((i.numpy() == wi.numpy()).all()) and bool(np.array_equal(v.numpy(), wv.numpy()))
        ok = ok and i[0, 0].item() == min(7, bounds[1] + 3)                      # the cross-shard duplicate resolved to the lower index
        sl = gal.my_slice(Ql)
        ok = ok and (sl.start == rank * Ql)
        out[rank] = ok
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,bounds,k", [(2, [0, 150, 300], 3), (3, [0, 100, 101, 260], 5), (2, [0, 2, 50], 4),
                                            (8, [0, 33, 33, 90, 91, 160, 200, 201, 260], 3)])   # 8 ranks, ragged, one empty shard
def test_sharded_equals_unsharded(world, bounds, k):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, bounds, k, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_world_size_one_never_touches_distributed():
    assert not torch.distributed.is_initialized()
    G = synth.normal(71, (120, 32))
    gal = ShardedGallery(torch.from_numpy(G), ops=OracleOps)
    v, i = gal.search(torch.from_numpy(G[:4].copy()), 2)
    assert i[:, 0].tolist() == [0, 1, 2, 3] and gal.world == 1 and gal.offsets == [0, 120]
