"""Row-sharded gallery across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, xGMI underneath).
The reference has no multi-GPU inference path (inference/inference.py:271 is single-device); this is
the one place the hot path has a real exchange step:

  1. every rank embeds its own images -> its gallery shard is *born* local, rows
     ``[offset[r], offset[r+1])`` of the global gallery; no collective during embedding
  2. queries are replicated with one all-gather of (Q_local, D) fp32 (1.5 MB at Q=256)
  3. each rank runs cosine + top-k over its shard (k <= 8: selected inside the GEMM epilogue, no score slab)
     -> (Q, k) {f32 score, i32 LOCAL index}
  4. ONE all-gather of the packed candidates (Q*k*8 bytes per rank: KBs, so latency- not link-bound; RCCL picks
     a direct one-hop exchange at this size, a ring would be 7 serial xGMI hops for nothing)
  5. every rank adds the shard offsets (a device tensor, no host sync) and merges world*k candidates per query
     (both inside ``mi355_merge_packed_topk``) with the same ordering rule (higher score, then LOWER global index) -> identical to the single-GPU result,
     bit for bit.

world_size == 1 never touches torch.distributed.
"""
from __future__ import annotations

import torch

from . import rank as _rank
from ._lib import MI355Error



class _HipOps:
    """Default compute backend: the HIP library.  (Tests inject a CPU backend to exercise the
    collective plumbing under gloo without a GPU; the product path is always this one.)"""

    @staticmethod
    def local_topk(queries, gallery_normalized, k, idx_offset, prepared=None):
        if prepared is not None and _rank.PreparedGallery.supports(queries.shape[0], k):
            return prepared.search(queries, k, idx_offset=idx_offset)
        return _rank.cosine_topk(queries, gallery_normalized, k, gallery_is_normalized=True, idx_offset=idx_offset)

    @staticmethod
    def prepare(gallery_normalized):
        return _rank.PreparedGallery(gallery_normalized) if gallery_normalized.shape[0] else None

    @staticmethod
    def pack(vals, idx, Q, k, device):
        return _rank.pack_candidates(vals, idx, Q, k, device)

    @staticmethod
    def merge_packed(packed, shard_offsets, k):
        return _rank.merge_packed_topk(packed, shard_offsets, k)

    @staticmethod
    def normalize(rows):
        return _rank.l2_normalize_rows(rows)


class ShardedGallery:
    def __init__(self, local_rows: torch.Tensor, group=None, ops=None, labels: torch.Tensor | None = None,
                 prepared: bool = False):
        self.ops = ops or _HipOps
        self.group = group
        dist = torch.distributed
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if local_rows.dim() != 2:
            raise MI355Error(f"local gallery shard must be (rows, dim), got {tuple(local_rows.shape)}")
        self.dim = local_rows.shape[1]
        self.device = local_rows.device
        self.local = self.ops.normalize(local_rows.float().contiguous()) if local_rows.shape[0] else local_rows.float()
        self.labels = labels
        # prepared=True: the shard is also kept as the cosine GEMM's bf16 planes (+6 B per element; same results bit for bit)
        self.prepared = self.ops.prepare(self.local) if (prepared and hasattr(self.ops, "prepare")) else None
        if self.world > 1:
            n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=self.device)
            allc = torch.empty(self.world, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, n, group=group)
            counts = allc.cpu().tolist()                     # ONE device->host copy for the whole table
        else:
            counts = [int(local_rows.shape[0])]
        if any(c >= 2 ** 31 - 128 for c in counts):
            raise MI355Error("a gallery shard must have fewer than 2^31 rows (candidates carry int32 local indices)")
        self.counts = counts
        self.offsets = [0]
        for c in counts:
            self.offsets.append(self.offsets[-1] + c)
        self.total_rows = self.offsets[-1]
        self._offsets_dev = torch.tensor(self.offsets[:-1], dtype=torch.int64, device=self.device)

    @property
    def offset(self) -> int:
        return self.offsets[self.rank]

    def _local_topk(self, queries, k):
        if self.prepared is not None:
            return self.ops.local_topk(queries, self.local, k, 0, prepared=self.prepared)
        return self.ops.local_topk(queries, self.local, k, 0)

    def _local_candidates(self, queries, k):
        """(Q, k, 2) int32: [..., 0] = the f32 score's bits, [..., 1] = LOCAL row index; a short (or empty) shard pads to
        exactly k slots with {-inf, -1} (one library kernel: mi355_pack_candidates)."""
        Q = queries.shape[0]
        kk = min(k, self.local.shape[0])
        v, i = self._local_topk(queries, kk) if kk > 0 else (None, None)
        return self.ops.pack(v, i, Q, k, self.device)

    def search(self, queries_local: torch.Tensor, k: int):
        """Top-k of every rank's queries against the WHOLE gallery.

        ``queries_local``: this rank's (Q_local, D) queries (same Q_local on every rank).
        Returns (values, global indices) for ALL world*Q_local queries, rank-major, on every rank."""
        if k < 1 or k > self.total_rows:
            raise MI355Error(f"selected index k out of range: k={k}, gallery rows={self.total_rows}")
        q = queries_local.float().contiguous()
        if self.world == 1:
            return self._local_topk(q, k)
        dist = torch.distributed
        Ql = q.shape[0]
        allq = torch.empty((self.world * Ql, self.dim), dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(allq, q, group=self.group)
        packed = self._local_candidates(allq, k)
        Q = allq.shape[0]
        allp = torch.empty((self.world * Q, k, 2), dtype=torch.int32, device=self.device)   # rank-major concat
        dist.all_gather_into_tensor(allp, packed, group=self.group)                             # the ONE candidate exchange
        # unpacking, the shard offsets and the merge of world * k candidates per query: one library call
        # (mi355_merge_packed_topk), no torch elementwise kernels on the rank stream
        return self.ops.merge_packed(allp.view(self.world, Q, k, 2), self._offsets_dev, k)

    def my_slice(self, Q_local: int) -> slice:
        """Rows of ``search``'s result that belong to this rank's own queries."""
        return slice(self.rank * Q_local, (self.rank + 1) * Q_local)
