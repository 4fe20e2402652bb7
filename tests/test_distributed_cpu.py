"""World-size-2 gloo test of the multi-GPU exchange step (ShardedSearch): shard by rows, local
exact top-k, all-gather (ids, scores, counts), k-way merge.  On CPU the local search and the merge
are test stand-ins (oracle / python); on the GPU box the same class runs the HIP kernels
(tests/test_search_gpu.py::test_sharded_equals_unsharded covers those)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_rows: int, q, k: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from cadence_rag_amd.sharded import ShardedSearch, shard_bounds
    from tests.helpers import cpu_merge_topk, unit_rows

    corpus = unit_rows(np.random.default_rng(77), n_rows)  # every rank derives the same corpus
    lo, hi = shard_bounds(n_rows, world, rank)

    def local(queries, kk):
        ids, sc, ct = oracle.exact_topk(queries.numpy(), corpus[lo:hi], kk, ids=np.arange(lo, hi), mode=oracle.F64)
        return torch.from_numpy(ids), torch.from_numpy(sc.astype(np.float32)), torch.from_numpy(ct)

    ids, sc, ct = ShardedSearch(None, local_search=local, merge=cpu_merge_topk).search(torch.from_numpy(q), k)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=ids.numpy(), sc=sc.numpy(), ct=ct.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_matches_single_scan(tmp_path):
    import oracle
    from tests.helpers import unit_rows
    n_rows, k = 1001, 10  # odd size: shards of 501 and 500 rows
    q = np.random.default_rng(5).standard_normal((4, 1024)).astype(np.float32)
    mp.spawn(_worker, args=(2, _free_port(), n_rows, q, k, str(tmp_path)), nprocs=2, join=True)
    corpus = unit_rows(np.random.default_rng(77), n_rows)
    want_ids, want_sc, want_ct = oracle.exact_topk(q, corpus, k, mode=oracle.F64)
    for rank in range(2):  # every rank holds the merged answer
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["ids"], want_ids)
        assert np.array_equal(got["ct"], want_ct)
        assert np.max(np.abs(got["sc"] - want_sc)) < 1e-6


def _packed_worker(rank: int, world: int, port: int, n_rows: int, q, k: int, out_dir: str, result_rank) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from cadence_rag_amd.dense_index import ResultRecord
    from cadence_rag_amd.sharded import ShardedSearch, shard_bounds
    from tests.helpers import cpu_merge_topk_packed, unit_rows

    corpus = unit_rows(np.random.default_rng(78), n_rows)
    lo, hi = shard_bounds(n_rows, world, rank)
    nbytes = ResultRecord.record_bytes(len(q), k)   # the C ABI's own layout arithmetic (no GPU needed)

    def local_into(queries, kk, rec):   # what crag_index_search_async does on the GPU: write ids | scores | counts
        ids, sc, ct = oracle.exact_topk(queries.numpy(), corpus[lo:hi], kk, ids=np.arange(lo, hi), mode=oracle.F64)
        rec.ids.copy_(torch.from_numpy(ids))
        rec.scores.copy_(torch.from_numpy(sc.astype(np.float32)))
        rec.counts.copy_(torch.from_numpy(ct))

    def merge_packed(gathered, w, nq, kk):
        return cpu_merge_topk_packed(gathered, w, nq, kk, nbytes)

    ss = ShardedSearch(None, local_into=local_into, merge_packed=merge_packed, result_rank=result_rank)
    for _ in range(2):   # twice: the second step reuses the gather buffer and the record laid over this rank's slot
        out = ss.search(torch.from_numpy(q), k)
    assert ss._rec.buf.data_ptr() == ss._gathered.data_ptr() + rank * nbytes    # in place: no separate record
    if out is None:
        assert result_rank is not None and rank != result_rank
    else:
        ids, sc, ct = out
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=ids.numpy(), sc=sc.numpy(), ct=ct.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_exchange_packed_result_records(tmp_path):
    """The HIP lane's exchange format under two ranks: real ResultRecord bytes (crag_result_record_bytes layout, an
    odd number of queries so that the record needs its padding, shards so small that counts < k) through ONE in-place
    all_gather_into_tensor and a CPU stand-in of merge_packed_kernel that parses the gathered bytes where they lie."""
    import oracle
    from tests.helpers import unit_rows
    n_rows, k = 13, 10   # shards of 7 and 6 rows: every per-shard list is shorter than k
    q = np.random.default_rng(6).standard_normal((5, 1024)).astype(np.float32)
    corpus = unit_rows(np.random.default_rng(78), n_rows)
    want_ids, want_sc, want_ct = oracle.exact_topk(q, corpus, k, mode=oracle.F64)
    for result_rank in (None, 1):
        out = tmp_path / f"rr{result_rank}"
        out.mkdir()
        mp.spawn(_packed_worker, args=(2, _free_port(), n_rows, q, k, str(out), result_rank), nprocs=2, join=True)
        holders = [0, 1] if result_rank is None else [result_rank]
        assert sorted(p.name for p in out.iterdir()) == [f"rank{r}.npz" for r in holders]
        for rank in holders:
            got = np.load(out / f"rank{rank}.npz")
            assert np.array_equal(got["ids"], want_ids)
            assert np.array_equal(got["ct"], want_ct) and int(want_ct.max()) == 10
            assert np.max(np.abs(got["sc"] - want_sc)) < 1e-6


def test_shard_bounds_partition_rows():
    from cadence_rag_amd.sharded import shard_bounds
    for n, w in ((10, 3), (1_000_000, 8), (5, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
