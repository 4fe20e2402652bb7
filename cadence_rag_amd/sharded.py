"""Multi-GPU dense search: one process per GPU, the corpus sharded by rows, per-shard exact top-k,
then the path's single exchange step — an all-gather of (ids, scores, counts)[nq, k] (12*nq*k
bytes per rank; RCCL over xGMI when the group's backend is "nccl") — and a k-way merge on every
rank (crag_merge_topk on the GPU).  No reference counterpart: the reference has no parallelism
(SURVEY.md 2); ordering rule as in crag_index_search (score desc, id asc).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .dense_index import DenseIndex, ResultRecord, merge_topk, merge_topk_packed


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row shard [lo, hi) of rank `rank`: sizes differ by at most one row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedSearch:
    def __init__(self, index: Optional[DenseIndex], group=None,
                 local_search: Optional[Callable] = None, merge: Optional[Callable] = None,
                 local_into: Optional[Callable] = None, merge_packed: Optional[Callable] = None,
                 result_rank: Optional[int] = None) -> None:
        """local_search(queries, k) -> (ids, scores, counts) tensors and merge(ids, scores, counts)
        -> (ids, scores, counts) default to the HIP paths; tests inject CPU stand-ins over gloo.
        local_into(queries, k, record) / merge_packed(gathered uint8, world, nq, k) -> (ids, scores, counts): stand-ins
        for the PACKED path (the one the HIP lane takes: the search writes a ResultRecord, one all-gather moves the
        records' bytes, the merge reads them in place) -- given both, that path runs over any backend.
        result_rank: only this rank needs the merged answer (the one that serves the request): the others take part
        in the exchange, skip the merge and return None.  Default: every rank merges (each holds the answer)."""
        self.index = index
        self.group = group
        self._local = local_search or self._hip_local
        self._merge = merge or self._hip_merge
        self._local_into = local_into
        self._merge_packed = merge_packed
        self.result_rank = result_rank

    def _hip_local(self, queries: torch.Tensor, k: int):
        if self.index is None:
            raise RuntimeError("ShardedSearch needs a DenseIndex (no CPU fallback)")
        nq = queries.shape[0]
        ids = torch.empty(nq, k, dtype=torch.int64, device=queries.device)
        sc = torch.empty(nq, k, dtype=torch.float32, device=queries.device)
        ct = torch.empty(nq, dtype=torch.int32, device=queries.device)
        self.index.search_async(queries, k, ids, sc, ct, stream=torch.cuda.current_stream().cuda_stream)
        return ids, sc, ct

    @staticmethod
    def _hip_merge(g_ids, g_sc, g_ct):
        _, nq, k = g_ids.shape
        ids = torch.empty(nq, k, dtype=torch.int64, device=g_ids.device)
        sc = torch.empty(nq, k, dtype=torch.float32, device=g_ids.device)
        ct = torch.empty(nq, dtype=torch.int32, device=g_ids.device)
        merge_topk(g_ids, g_sc, g_ct, ids, sc, ct, stream=torch.cuda.current_stream().cuda_stream)
        return ids, sc, ct

    def _search_packed(self, queries: torch.Tensor, k: int):
        """The search writes into this rank's OWN SLOT of the gather buffer (a ResultRecord view of it), ONE in-place
        all-gather moves the records, the merge reads the gathered bytes where they lie.  [Round 3 gathered from a
        separate record: one more device copy of it per step inside the collective.]"""
        nq = int(queries.shape[0])
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        key = (nq, k, world, str(queries.device))
        if getattr(self, "_packed_key", None) != key:
            nbytes = ResultRecord.record_bytes(nq, k)
            self._gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device=queries.device)
            self._rec = ResultRecord(nq, k, queries.device, buf=self._gathered[rank * nbytes:(rank + 1) * nbytes])
            self._packed_key = key
        rec = self._rec
        hip = self._local_into is None
        stream = torch.cuda.current_stream().cuda_stream if hip else 0
        if hip:
            self.index.search_async(queries, k, rec.ids, rec.scores, rec.counts, stream=stream)
        else:
            self._local_into(queries, k, rec)
        dist.all_gather_into_tensor(self._gathered, rec.buf, group=self.group)
        if self.result_rank is not None and rank != self.result_rank:
            return None
        if not hip:
            return self._merge_packed(self._gathered, world, nq, k)
        ids = torch.empty(nq, k, dtype=torch.int64, device=queries.device)
        sc = torch.empty(nq, k, dtype=torch.float32, device=queries.device)
        ct = torch.empty(nq, dtype=torch.int32, device=queries.device)
        merge_topk_packed(self._gathered, world, nq, k, ids, sc, ct, stream=stream)
        return ids, sc, ct

    def search(self, queries: torch.Tensor, k: int, force_exchange: bool = False):
        """force_exchange: run the all-gather + merge even in a group of one rank (how the RCCL path is
        exercised on a single-GPU box)."""
        hip_path = (self.index is not None and self._local == self._hip_local and self._merge == self._hip_merge
                    and dist.is_initialized())
        packed_standins = self._local_into is not None and self._merge_packed is not None and dist.is_initialized()
        if (hip_path or packed_standins) and (dist.get_world_size(self.group) > 1 or force_exchange):
            return self._search_packed(queries, k)
        ids, sc, ct = self._local(queries, k)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return ids, sc, ct
        nq, k = ids.shape
        # concatenated layout (dim 0 = rank-major): the form both NCCL/RCCL and gloo accept
        g_ids = torch.empty((world * nq, k), dtype=ids.dtype, device=ids.device)
        g_sc = torch.empty((world * nq, k), dtype=sc.dtype, device=sc.device)
        g_ct = torch.empty((world * nq,), dtype=ct.dtype, device=ct.device)
        dist.all_gather_into_tensor(g_ids, ids.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_sc, sc.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_ct, ct.contiguous(), group=self.group)
        return self._merge(g_ids.view(world, nq, k), g_sc.view(world, nq, k), g_ct.view(world, nq))
