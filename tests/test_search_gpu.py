"""Parity of the HIP dense-search path against the CPU oracle and the committed golden fixtures,
through the C ABI (DenseIndex -> libcrag_dense.so).  Bar (BASELINE.json): identical top-k id order,
|dscore| <= 1e-4 vs the fp64 exact scan, recall@10 = 1.0."""
from pathlib import Path

import numpy as np
import pytest

import oracle
from cadence_rag_amd import retrieve as rt
from cadence_rag_amd.dense_index import DenseIndex, merge_topk
from tests.helpers import assert_topk_matches, cpu_merge_topk, unit_rows

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
TOL = 1e-4  # BASELINE.json: cosine scores within 1e-4 (fp32)


def _check(corpus, queries, k, *, mask=None, ids=None, index=None):
    ix = index or DenseIndex(corpus.shape[1], capacity=max(len(corpus), 1))
    try:
        if index is None and len(corpus):
            ix.add(corpus, ids=ids)
        packed = None if mask is None else DenseIndex.pack_mask(mask)
        got = ix.search(queries, k, row_mask=packed)
        omask = None if mask is None else np.packbits(mask, axis=-1, bitorder="little")
        want = oracle.exact_topk(queries, corpus, k, ids=ids, mask=omask, mode=oracle.F64)
        assert_topk_matches(*got, *want, tol=TOL)
        return got
    finally:
        if index is None:
            ix.close()


def test_golden_4096(gpu):
    g = np.load(GOLD / "dense_scan_4096.npz")
    rng = np.random.default_rng(int(g["seed"]))
    a = rng.standard_normal((int(g["n"]), 1024))
    corpus = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    with DenseIndex(1024, capacity=4096) as ix:
        ix.add(corpus)
        ids, scores, counts = ix.search(g["queries"], int(g["k"]))
    assert np.array_equal(ids, g["ids"].astype(np.int64))  # bit-exact order on the pinned fixture
    assert np.max(np.abs(scores - g["scores"])) <= TOL
    assert np.all(counts == int(g["k"]))


def test_golden_edge_cases(gpu):
    g = np.load(GOLD / "dense_scan_257_edge.npz")
    k = int(g["k"])
    with DenseIndex(1024, capacity=257) as ix:
        ix.add(g["corpus"])
        ids, scores, counts = ix.search(g["queries"], k)
        assert np.array_equal(ids, g["ids"].astype(np.int64))  # ties by ascending id, NaN/zero rows absent
        assert np.array_equal(counts, g["counts"])
        assert np.max(np.abs(scores - g["scores"])) <= TOL
        ids, scores, counts = ix.search(g["queries"], k, row_mask=DenseIndex.pack_mask(g["mask"]))
        assert np.array_equal(ids, g["masked_ids"].astype(np.int64))
        assert np.array_equal(counts, g["masked_counts"])
        assert ix.count_eligible() == 257 - 2  # zero row + NaN row
        assert ix.count_eligible(DenseIndex.pack_mask(g["mask"][0])) == int(
            (g["mask"][0] & ~np.isin(np.arange(257), [5, 100])).sum())


@pytest.mark.parametrize("n,nq,k", [
    (1, 1, 1), (7, 2, 10), (31, 2, 10), (32, 3, 32), (33, 1, 5), (257, 1, 5), (1000, 3, 10),
    (4096, 32, 10), (5000, 33, 33), (20000, 33, 50), (9999, 5, 64), (30000, 64, 100), (12345, 7, 128),
])
def test_random_corpora_match_oracle(gpu, n, nq, k):
    rng = np.random.default_rng(n * 7919 + nq * 31 + k)
    _check(unit_rows(rng, n), rng.standard_normal((nq, 1024)).astype(np.float32), k)


def test_unnormalised_rows_and_queries(gpu):
    rng = np.random.default_rng(99)
    corpus = (rng.standard_normal((3000, 1024)) * rng.uniform(0.01, 50, (3000, 1))).astype(np.float32)
    q = (rng.standard_normal((4, 1024)) * 1e3).astype(np.float32)
    _check(corpus, q, 20)


@pytest.mark.parametrize("frac,per_query", [(0.5, False), (0.1, True), (0.001, True), (0.0, False), (1.0, True)])
def test_row_masks(gpu, frac, per_query):
    rng = np.random.default_rng(int(frac * 1000) + per_query)
    corpus = unit_rows(rng, 5000)
    q = rng.standard_normal((8, 1024)).astype(np.float32)
    mask = rng.random((8, 5000) if per_query else (5000,)) < frac
    _check(corpus, q, 10, mask=mask)


@pytest.mark.parametrize("nq,k,frac", [(9, 100, 0.01), (40, 64, 0.1), (64, 128, 0.5), (33, 40, 0.002)])
def test_row_masks_large_k(gpu, nq, k, frac):
    """k > 32 runs the LDS-list kernels (32 / 64 queries per pass); sparse masks leave fewer than k rows."""
    rng = np.random.default_rng(nq * 1000 + k)
    corpus = unit_rows(rng, 6000)
    q = rng.standard_normal((nq, 1024)).astype(np.float32)
    mask = rng.random((nq, 6000)) < frac
    _check(corpus, q, k, mask=mask)


@pytest.mark.parametrize("nq,k", [(3, 40), (64, 100)])
def test_exact_ties_large_k(gpu, nq, k):
    rng = np.random.default_rng(5 + nq)
    corpus = unit_rows(rng, 4000)
    corpus[2000:2060] = corpus[7]  # 60 identical rows + the original
    q = rng.standard_normal((nq, 1024)).astype(np.float32)
    q[0] = corpus[7] * 3.0
    ids, scores, counts = _check(corpus, q, k)
    want = ([7] + list(range(2000, 2060)))[:k]
    assert list(ids[0, :len(want)]) == want


def test_exact_ties_break_by_ascending_position(gpu):
    rng = np.random.default_rng(3)
    corpus = unit_rows(rng, 3000)
    corpus[1500] = corpus[3]
    corpus[2999] = corpus[3]
    corpus[10:20] = corpus[10]  # ten identical rows
    q = np.stack([corpus[3] * 2.0, corpus[10], rng.standard_normal(1024).astype(np.float32)])
    ids, scores, counts = _check(corpus, q, 20)
    assert list(ids[0, :3]) == [3, 1500, 2999]
    assert list(ids[1, :10]) == list(range(10, 20))


def test_zero_nan_inf_rows_and_bad_queries(gpu):
    rng = np.random.default_rng(4)
    corpus = unit_rows(rng, 1000)
    corpus[0] = 0.0
    corpus[1, 5] = np.nan
    corpus[2, 7] = np.inf
    q = rng.standard_normal((3, 1024)).astype(np.float32)
    q[1] = 0.0        # zero query: pgvector gives NaN distances -> nothing eligible
    q[2, 0] = np.nan  # NaN query
    ids, scores, counts = _check(corpus, q, 10)
    assert counts.tolist() == [10, 0, 0]
    assert not np.isin(ids[0], [0, 1, 2]).any()


def test_small_dims_are_zero_padded(gpu):
    for dim in (4, 100, 256, 1000):
        rng = np.random.default_rng(dim)
        corpus = rng.standard_normal((777, dim)).astype(np.float32)
        q = rng.standard_normal((5, dim)).astype(np.float32)
        _check(corpus, q, 10)


def test_external_ids_incremental_add_update_and_roundtrip(gpu):
    rng = np.random.default_rng(8)
    a, b = unit_rows(rng, 1000), unit_rows(rng, 777)
    ids_a = np.arange(10_000, 11_000, dtype=np.int64)
    ids_b = np.arange(50_000_000_000, 50_000_000_777, dtype=np.int64)  # > 2^32: ids are 64-bit
    q = rng.standard_normal((4, 1024)).astype(np.float32)
    with DenseIndex(1024, capacity=2000) as ix:
        ix.add(a, ids=ids_a)
        ix.add(b, ids=ids_b)
        assert len(ix) == 1777
        both, ids = np.concatenate([a, b]), np.concatenate([ids_a, ids_b])
        _check(both, q, 25, ids=ids, index=ix)
        rows, got_ids = ix.get_rows(990, 20)  # straddles the two adds and a 32-row tile boundary
        assert np.array_equal(rows, both[990:1010]) and np.array_equal(got_ids, ids[990:1010])
        fresh = unit_rows(rng, 5)
        ix.update(998, fresh)  # re-embed in place
        both[998:1003] = fresh
        _check(both, q, 25, ids=ids, index=ix)
        with pytest.raises(Exception, match="capacity exceeded"):
            ix.add(unit_rows(rng, 300))


def test_device_pointers_and_async_entry(gpu):
    import torch
    rng = np.random.default_rng(12)
    corpus, q = unit_rows(rng, 6000), rng.standard_normal((9, 1024)).astype(np.float32)
    dev = torch.device("cuda", 0)
    with DenseIndex(1024, capacity=6000) as ix:
        ix.add(torch.from_numpy(corpus).to(dev))
        dq = torch.from_numpy(q).to(dev)
        oi = torch.empty(9, 10, dtype=torch.int64, device=dev)
        osc = torch.empty(9, 10, dtype=torch.float32, device=dev)
        oc = torch.empty(9, dtype=torch.int32, device=dev)
        for _ in range(3):  # consecutive passes alternate scan direction; results must not change
            ix.search_async(dq, 10, oi, osc, oc, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want = oracle.exact_topk(q, corpus, 10, mode=oracle.F64)
            assert_topk_matches(oi.cpu().numpy(), osc.cpu().numpy(), oc.cpu().numpy(), *want, tol=TOL)


def test_argument_validation_through_the_abi(gpu):
    rng = np.random.default_rng(1)
    with DenseIndex(1024, capacity=100) as ix:
        ix.add(unit_rows(rng, 100))
        q = rng.standard_normal((1, 1024)).astype(np.float32)
        for bad_k in (0, 129, -3):
            with pytest.raises(Exception, match="k must be in"):
                ix.search(q, bad_k)
        with pytest.raises(ValueError):
            ix.search(rng.standard_normal((1, 512)).astype(np.float32), 5)


def test_merge_topk_kernel_matches_cpu_merge(gpu):
    import torch
    rng = np.random.default_rng(21)
    r, nq, k = 8, 13, 10
    sc = np.sort(rng.standard_normal((r, nq, k)).astype(np.float32), axis=-1)[..., ::-1].copy()
    sc[3, :, 2] = sc[0, :, 1]  # cross-shard exact ties: lower id must win
    ids = rng.permutation(r * nq * k).reshape(r, nq, k).astype(np.int64)
    ct = rng.integers(0, k + 1, size=(r, nq)).astype(np.int32)
    ct[:, 0] = 0  # a query with no eligible row anywhere
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x).to(dev) for x in (ids, sc, ct)]
    oi = torch.empty(nq, k, dtype=torch.int64, device=dev)
    osc = torch.empty(nq, k, dtype=torch.float32, device=dev)
    oc = torch.empty(nq, dtype=torch.int32, device=dev)
    merge_topk(*t, oi, osc, oc)
    torch.cuda.synchronize()
    wi, ws, wc = cpu_merge_topk(*[torch.from_numpy(x) for x in (ids, sc, ct)])
    assert torch.equal(oc.cpu(), wc)
    assert torch.equal(oi.cpu(), wi)
    assert torch.equal(osc.cpu().nan_to_num(nan=7.0), ws.nan_to_num(nan=7.0))


def test_sharded_equals_unsharded(gpu):
    """Two shards on one GPU + the merge kernel == one index over the whole corpus."""
    import torch
    rng = np.random.default_rng(33)
    corpus, q = unit_rows(rng, 9000), rng.standard_normal((6, 1024)).astype(np.float32)
    k, dev = 10, torch.device("cuda", 0)
    parts = []
    for lo, hi in ((0, 4500), (4500, 9000)):
        with DenseIndex(1024, capacity=4500) as ix:
            ix.add(corpus[lo:hi], ids=np.arange(lo, hi))
            parts.append(ix.search(q, k))
    g = [torch.from_numpy(np.stack([p[i] for p in parts])).to(dev) for i in range(3)]
    oi = torch.empty(6, k, dtype=torch.int64, device=dev)
    osc = torch.empty(6, k, dtype=torch.float32, device=dev)
    oc = torch.empty(6, dtype=torch.int32, device=dev)
    merge_topk(*g, oi, osc, oc)
    torch.cuda.synchronize()
    want = oracle.exact_topk(q, corpus, k, mode=oracle.F64)
    assert_topk_matches(oi.cpu().numpy(), osc.cpu().numpy(), oc.cpu().numpy(), *want, tol=TOL)


def test_full_size_properties_100k(gpu):
    """BASELINE configs[1] size: properties that need no full oracle pass — planted neighbours are
    found, scores sorted, repeatable, and a sampled oracle check gives recall@10 = 1.0."""
    import torch
    g = torch.Generator().manual_seed(1234)
    c = torch.randn(100_000, 1024, generator=g)
    c /= c.norm(dim=1, keepdim=True)
    corpus = c.numpy()
    rng = np.random.default_rng(4321)
    planted_rows = rng.choice(100_000, size=32, replace=False)
    q = corpus[planted_rows] + 0.05 * rng.standard_normal((32, 1024)).astype(np.float32) / 32.0
    with DenseIndex(1024, capacity=100_000) as ix:
        ix.add(corpus)
        ids, scores, counts = ix.search(q, 10)
        ids2, scores2, _ = ix.search(q, 10)  # the reversed pass
        assert np.array_equal(ids, ids2) and np.array_equal(scores, scores2)
        assert np.array_equal(ids[:, 0], planted_rows)
        assert np.all(np.diff(scores, axis=1) <= 0) and np.all(counts == 10)
        big_ids, big_scores, _ = ix.search(q, 100)
        assert np.array_equal(big_ids[:, :10], ids)  # top-10 is a prefix of top-100 (two kernels)
        want = oracle.exact_topk(q[:8], corpus, 10, mode=oracle.F64, fast=True)
        assert_topk_matches(ids[:8], scores[:8], counts[:8], *want, tol=TOL)
        recall = np.mean([len(set(ids[i]) & set(want[0][i])) / 10 for i in range(8)])
        assert recall == 1.0


def test_dense_table_lane_rows_and_filters(gpu):
    """_fetch_chunks_dense / _fetch_artifacts_dense row shapes and filter semantics."""
    from datetime import datetime, timedelta, timezone
    from uuid import UUID
    rng = np.random.default_rng(5)
    n = 400
    vecs = unit_rows(rng, n)
    calls = [UUID(int=i % 4 + 1) for i in range(n)]
    t0 = datetime(2026, 1, 1, tzinfo=timezone.utc)
    started = [(t0 + timedelta(days=i % 4)).replace(tzinfo=None) for i in range(n)]
    table = rt.DenseTable("chunks", "chunk_id", dim=1024, capacity=n)
    try:
        table.add(vecs, {"chunk_id": list(range(100, 100 + n)), "call_id": calls,
                         "speaker": ["S"] * n, "start_ts_ms": list(range(n)), "end_ts_ms": list(range(1, n + 1)),
                         "text": [f"chunk {i}" for i in range(n)]},
                  call_started_at=started,
                  call_tags={UUID(int=1): ["alpha"], UUID(int=2): ["beta"], UUID(int=3): [], UUID(int=4): ["alpha", "x"]})
        q = vecs[7] * 3.0
        rows = rt._fetch_chunks_dense(table, rt._vector_literal(q.tolist()), None, None, "exact", 5)
        assert list(rows[0]) == ["chunk_id", "call_id", "speaker", "start_ts_ms", "end_ts_ms", "text", "score"]
        assert rows[0]["chunk_id"] == 107 and rows[0]["text"] == "chunk 7" and rows[0]["score"] == pytest.approx(1.0, abs=1e-6)
        assert [r["score"] for r in rows] == sorted((r["score"] for r in rows), reverse=True)
        # call_ids scope
        f = rt.RetrieveFilters(call_ids=[UUID(int=2)])
        rows = rt._fetch_chunks_dense(table, q, f, f.call_ids, "exact", 50)
        assert len(rows) == 50 and all(r["call_id"] == UUID(int=2) for r in rows)
        assert rt._estimate_dense_candidates(table, "chunks", f, f.call_ids) == 100
        # date range + tags
        f = rt.RetrieveFilters(date_from=t0 + timedelta(days=1), date_to=t0 + timedelta(days=2), call_tags=["beta"])
        rows = rt._fetch_chunks_dense(table, q, f, None, "ann", 1000)
        assert len(rows) == 100 and {r["call_id"] for r in rows} == {UUID(int=2)}
        want = oracle.exact_topk(q[None], vecs, 128, mask=np.packbits(np.array([c == UUID(int=2) for c in calls]),
                                                                       bitorder="little"), mode=oracle.F64)
        assert [r["chunk_id"] - 100 for r in rows] == want[0][0, :100].tolist()
        # empty scope -> no rows, planner says exact
        assert rt._fetch_chunks_dense(table, q, rt.RetrieveFilters(external_id="zz"), [], "exact", 10) == []
        assert rt._estimate_dense_candidates(table, "chunks", rt.RetrieveFilters(external_id="zz"), []) == 0
    finally:
        table.close()


def test_packed_result_records_merge(gpu):
    """The one-collective exchange form: two shards write into ResultRecords, the records are
    concatenated as an all-gather would, crag_merge_topk_packed gives the unsharded answer."""
    import torch
    from cadence_rag_amd.dense_index import ResultRecord, merge_topk_packed
    rng = np.random.default_rng(44)
    corpus, q = unit_rows(rng, 7001), rng.standard_normal((5, 1024)).astype(np.float32)
    k, dev = 7, torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    recs = []
    for lo, hi in ((0, 3500), (3500, 7001)):
        with DenseIndex(1024, capacity=hi - lo) as ix:
            ix.add(corpus[lo:hi], ids=np.arange(lo, hi))
            rec = ResultRecord(5, k, dev)
            ix.search_async(dq, k, rec.ids, rec.scores, rec.counts, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            recs.append(rec.buf.clone())
    gathered = torch.cat(recs)
    oi = torch.empty(5, k, dtype=torch.int64, device=dev)
    osc = torch.empty(5, k, dtype=torch.float32, device=dev)
    oc = torch.empty(5, dtype=torch.int32, device=dev)
    merge_topk_packed(gathered, 2, 5, k, oi, osc, oc)
    torch.cuda.synchronize()
    want = oracle.exact_topk(q, corpus, k, mode=oracle.F64)
    assert_topk_matches(oi.cpu().numpy(), osc.cpu().numpy(), oc.cpu().numpy(), *want, tol=TOL)


def test_gpu_rrf_fusion_matches_reference_rrf_merge(gpu):
    """crag_rrf_fuse == _rrf_merge: the reference's golden cases (ties included) and random lanes."""
    import json
    import torch
    from cadence_rag_amd.fusion import rrf_fuse
    dev = torch.device("cuda", 0)
    gold = json.loads((GOLD / "reference_host_logic.json").read_text())["rrf_merge"]
    rng = np.random.default_rng(17)
    cases = [[(name, ids) for name, ids in c["lanes"]] for c in gold]
    for _ in range(20):  # bm25 50 / tech 50 / dense 100 shaped lanes with heavy overlap
        pool = rng.permutation(300)
        cases.append([("bm25", rng.choice(pool[:120], size=rng.integers(0, 51), replace=False).tolist()),
                      ("tech_tokens", rng.choice(pool[:150], size=rng.integers(0, 51), replace=False).tolist()),
                      ("dense", rng.choice(pool[:200], size=rng.integers(1, 101), replace=False).tolist())])
    n_lanes = 3
    widths = [max(max((len(dict(c).get(n, [])) for c in cases), default=1), 1) for n in ("bm25", "tech_tokens", "dense")]
    lanes = []
    for li, name in enumerate(("bm25", "tech_tokens", "dense")):
        ids = np.full((len(cases), widths[li]), -7, dtype=np.int64)
        cnt = np.zeros(len(cases), dtype=np.int32)
        for q, c in enumerate(cases):
            v = dict(c).get(name, [])
            ids[q, :len(v)] = v
            cnt[q] = len(v)
        lanes.append((torch.from_numpy(ids).to(dev), torch.from_numpy(cnt).to(dev)))
    out = rrf_fuse(lanes, out_k=256)
    torch.cuda.synchronize()
    ids, scores, masks, counts = (out[k].cpu().numpy() for k in ("ids", "scores", "lanes", "counts"))
    names = ("bm25", "tech_tokens", "dense")
    for q, c in enumerate(cases):
        # the reference iterates lanes in dict order; a lane missing from a golden case is simply absent
        lane_rows = {name: [{"id": i} for i in dict(c).get(name, [])] for name in names if name in dict(c)}
        want = rt._rrf_merge(lane_rows, "id")
        assert counts[q] == len(want)
        assert ids[q, :counts[q]].tolist() == [w[0]["id"] for w in want]
        assert scores[q, :counts[q]].tolist() == [w[2] for w in want]  # bit-identical fp64
        for pos, w in enumerate(want):
            assert {names[b] for b in range(n_lanes) if masks[q, pos] >> b & 1} == w[1]
        assert np.all(ids[q, counts[q]:] == -1)


def test_gpu_tech_token_lane_matches_sql_semantics(gpu):
    """tech_tokens && :tokens ORDER BY call_started_at DESC, id ASC LIMIT k  (retrieve.py:183-242)."""
    import torch
    from cadence_rag_amd.fusion import TechTokenIndex
    from cadence_rag_amd.tech_tokens import extract_tech_tokens
    rng = np.random.default_rng(23)
    vocab = ["ECONNRESET", "ABC-123", "v1.2.3", "HTTP 502", "BOM", "Dell", "vs", "10.25.0.50", "/etc/hosts", "Azure",
             "econnreset"]  # exact match: case differs -> no overlap
    n = 5000
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 4), replace=False)) for _ in range(n)]
    ids = rng.permutation(n).astype(np.int64) + 1000
    base = np.datetime64("2026-01-01T00:00:00", "us")
    started = base + rng.integers(0, 50, size=n).astype("timedelta64[D]")  # many ties -> id ASC decides
    dev = torch.device("cuda", 0)
    index = TechTokenIndex(row_tokens, ids, started, dev)
    queries = ["Where did we discuss ECONNRESET in api-gateway? ticket ABC-123 v1.2.3", "Dell vs Azure BOM",
               "nothing technical here", "HTTP 502 from 10.25.0.50"]
    qtoks = [extract_tech_tokens(q) for q in queries]
    elig = rng.random(n) < 0.7
    mask = torch.from_numpy(DenseIndex.pack_mask(elig)).to(dev)
    for use_mask in (False, True):
        out_ids, out_ct = index.search(qtoks, 50, row_mask=mask if use_mask else None)
        torch.cuda.synchronize()
        got_ids, got_ct = out_ids.cpu().numpy(), out_ct.cpu().numpy()
        for qi, toks in enumerate(qtoks):
            hits = [i for i in range(n) if set(toks) & set(row_tokens[i]) and (elig[i] or not use_mask)]
            hits.sort(key=lambda i: (-started[i].astype(np.int64), ids[i]))
            want = [int(ids[i]) for i in hits[:50]]
            assert got_ct[qi] == len(want)
            assert got_ids[qi, :len(want)].tolist() == want
            assert np.all(got_ids[qi, len(want):] == -1)
    assert qtoks[2] == [] and got_ct[2] == 0


def test_retrieve_evidence_over_gpu_backend_matches_cpu_composition(gpu, monkeypatch):
    """The /retrieve entry point over GpuRetrieveBackend (dense + exact-token lanes on the GPU, BM25 rows
    injected) against the same orchestration fed by the CPU oracle and a Python token filter."""
    from datetime import datetime, timedelta
    from uuid import UUID

    from cadence_rag_amd import embeddings
    from cadence_rag_amd.tech_tokens import extract_tech_tokens
    rng = np.random.default_rng(77)
    vocab = ["ERR-4521", "gpu-17", "v2.4.1", "OPS-99", "10.0.0.7", "ECONNRESET"]
    t0 = datetime(2026, 3, 1)
    calls = [{"call_id": UUID(int=i + 1), "external_id": f"ext-{i}", "external_source": "zoom"} for i in range(6)]

    def make(name, id_field, n, select_extra):
        vecs = unit_rows(rng, n)
        call = [calls[i % 6]["call_id"] for i in range(n)]
        cols = {id_field: [1000 + i for i in range(n)], "call_id": call}
        cols.update(select_extra(n))
        started = [t0 + timedelta(days=(i % 6)) for i in range(n)]
        toks = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
        table = rt.DenseTable(name, id_field, dim=1024, capacity=n)
        table.add(vecs, cols, call_started_at=started, call_tags={c["call_id"]: ["t%d" % (i % 2)] for i, c in enumerate(calls)})
        return table, vecs, toks, started

    chunks, cvec, ctok, cstart = make("chunks", "chunk_id", 700, lambda n: {
        "speaker": ["S%d" % (i % 3) for i in range(n)], "start_ts_ms": [i * 10 for i in range(n)],
        "end_ts_ms": [i * 10 + 9 for i in range(n)], "text": ["chunk text %d " % i * 3 for i in range(n)]})
    arts, avec, atok, astart = make("artifact_chunks", "artifact_chunk_id", 90, lambda n: {
        "artifact_id": [i // 3 for i in range(n)], "kind": ["summary"] * n, "content": ["artifact %d" % i for i in range(n)]})
    bm25_c = [{k: chunks.columns[k][p] for k in rt.CHUNK_SELECT} | {"score": 5.0 - j} for j, p in enumerate([3, 77, 500])]
    qvec = (cvec[77] + 0.5 * avec[5]).astype(np.float32)
    monkeypatch.setattr(embeddings, "embeddings_enabled", lambda: True)
    monkeypatch.setattr(embeddings, "embed_texts",
                        lambda texts: embeddings.EmbeddingResult(vectors=[qvec.tolist() for _ in texts], model="m"))

    class CpuBackend(rt.RetrieveBackend):
        def resolve_call_ids(self, filters): return rt._resolve_call_ids(calls, filters)
        def fetch_chunks_bm25(self, q, f, c, k): return [dict(r) for r in bm25_c][:k]

        def _mask(self, table, f, c):
            m = table.filter_mask(f, c)
            return np.ones(len(table), bool) if m is None else m

        def _tech(self, table, toks_by_row, started, select, t, f, c, k):
            if not t:
                return []
            m = self._mask(table, f, c)
            ids = table.columns[table.id_field]
            hits = [i for i in range(len(table)) if m[i] and set(t) & set(toks_by_row[i])]
            hits.sort(key=lambda i: (-np.datetime64(started[i], "us").astype(np.int64), ids[i]))
            return [{col: table.columns[col][i] for col in select} for i in hits[:k]]

        def fetch_chunks_tech(self, t, f, c, k): return self._tech(chunks, ctok, cstart, rt.CHUNK_SELECT, t, f, c, k)
        def fetch_artifacts_tech(self, t, f, c, k): return self._tech(arts, atok, astart, rt.ARTIFACT_SELECT, t, f, c, k)

        def estimate_dense_candidates(self, name, f, c):
            return int(self._mask(chunks if name == "chunks" else arts, f, c).sum())

        def _dense(self, table, vecs, select, e, f, c, k):
            m = self._mask(table, f, c)
            ids, sc, ct = oracle.exact_topk(rt._parse_vector(e)[None], vecs, k,
                                            mask=np.packbits(m, bitorder="little"), mode=oracle.F64)
            return [{col: table.columns[col][int(p)] for col in select} | {"score": float(s)}
                    for p, s in zip(ids[0, :ct[0]], sc[0, :ct[0]])]

        def fetch_chunks_dense(self, e, f, c, mode, k): return self._dense(chunks, cvec, rt.CHUNK_SELECT, e, f, c, k)
        def fetch_artifacts_dense(self, e, f, c, mode, k): return self._dense(arts, avec, rt.ARTIFACT_SELECT, e, f, c, k)

    try:
        gpu_be = rt.GpuRetrieveBackend(chunks, arts, calls=calls, bm25_chunks=lambda q, f, c, k: bm25_c[:k],
                                       tech_chunks=chunks.build_tech_lane(ctok), tech_artifacts=arts.build_tech_lane(atok))
        cases = [
            rt.RetrieveRequest(query="why ERR-4521 on gpu-17 after v2.4.1?", debug=True),
            rt.RetrieveRequest(query="why ERR-4521 on gpu-17 after v2.4.1?", return_style="ids_only"),
            rt.RetrieveRequest(query="OPS-99 status", debug=True, return_style="ids_only",
                               filters=rt.RetrieveFilters(call_ids=[calls[1]["call_id"], calls[4]["call_id"]])),
            rt.RetrieveRequest(query="ECONNRESET 10.0.0.7", debug=True,
                               filters=rt.RetrieveFilters(date_from=t0 + timedelta(days=2), call_tags=["t1"]),
                               budget=rt.Budget(max_evidence_items=5, max_total_chars=300)),
            rt.RetrieveRequest(query="no technical tokens at all", debug=True, return_style="ids_only"),
        ]
        for req in cases:
            got = rt.retrieve_evidence(req, gpu_be)
            want = rt.retrieve_evidence(req, CpuBackend())
            got.pop("query_id"); want.pop("query_id")
            # dense scores agree to fp32 rounding, everything else exactly
            for side in (got, want):
                for lanes in side.get("debug", {}).get("lanes", {}).values():
                    for row in lanes.get("dense", []):
                        row["score"] = round(row["score"], 5)
            assert got == want, req
            if req.return_style != "ids_only":
                assert got["notes"]["retrieval"]["lanes"] == {"bm25": True, "tech_tokens": True, "dense": True}
    finally:
        chunks.close(); arts.close()


@pytest.mark.parametrize("overlap", [True, False])
def test_hybrid_searcher_batch_matches_python_rrf(gpu, overlap):
    """HybridSearcher (dense + token lane + given BM25 -> RRF, all on the GPU) against the oracle's dense
    order, a Python token filter and the host mirror of _rrf_merge, incl. a per-query row mask; with the token lane
    on a side stream beside the dense scan (the default) and with the lanes in series."""
    import torch
    from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex
    rng = np.random.default_rng(321)
    n, nq = 3000, 37
    corpus = unit_rows(rng, n)
    vocab = ["T-%d" % i for i in range(40)]
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
    ids = np.arange(n, dtype=np.int64) + 500
    started = np.datetime64("2026-02-01", "us") + rng.integers(0, 30, size=n).astype("timedelta64[D]")
    qvec = rng.standard_normal((nq, 1024)).astype(np.float32)
    qtoks = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(nq)]
    bm25 = np.stack([rng.choice(ids, size=20, replace=False) for _ in range(nq)])
    bm25_ct = rng.integers(0, 21, size=nq).astype(np.int32)
    elig = rng.random((nq, n)) < 0.6
    dev = torch.device("cuda", 0)
    index = DenseIndex(1024, capacity=n)
    try:
        index.add(corpus, ids=ids)
        tech = TechTokenIndex(row_tokens, ids, started, dev)
        hs = HybridSearcher(index, tech, dense_k=40, tech_k=15, overlap_lanes=overlap)
        packed = DenseIndex.pack_mask(elig)
        d_mask = torch.from_numpy(packed).to(dev)
        out = hs.search(torch.from_numpy(qvec).to(dev), qtoks,
                        (torch.from_numpy(bm25).to(dev), torch.from_numpy(bm25_ct).to(dev)),
                        row_mask=d_mask, mask_stride=packed.shape[1])
        torch.cuda.synchronize()
        got_ids, got_ct = out["ids"].cpu().numpy(), out["counts"].cpu().numpy()
        got_sc, got_lanes = out["scores"].cpu().numpy(), out["lanes"].cpu().numpy()
        want_dense = oracle.exact_topk(qvec, corpus, 40, mask=np.packbits(elig, axis=-1, bitorder="little"), mode=oracle.F64)
        for qi in range(nq):
            dense_rows = [{"id": int(ids[p])} for p in want_dense[0][qi, :want_dense[2][qi]]]
            hits = [i for i in range(n) if elig[qi, i] and set(qtoks[qi]) & set(row_tokens[i])]
            hits.sort(key=lambda i: (-started[i].astype(np.int64), ids[i]))
            tech_rows = [{"id": int(ids[i])} for i in hits[:15]]
            bm_rows = [{"id": int(v)} for v in bm25[qi, :bm25_ct[qi]]]
            want = rt._rrf_merge({"bm25": bm_rows, "tech_tokens": tech_rows, "dense": dense_rows}, "id")
            assert got_ct[qi] == len(want)
            assert got_ids[qi, :len(want)].tolist() == [row["id"] for row, _, _ in want]
            assert np.array_equal(got_sc[qi, :len(want)], np.array([s for _, _, s in want]))
            lane_bits = {"bm25": 1, "tech_tokens": 2, "dense": 4}
            assert got_lanes[qi, :len(want)].tolist() == [sum(lane_bits[l] for l in lanes) for _, lanes, _ in want]
    finally:
        index.close()


@pytest.mark.parametrize("n,nq,k", [(20000, 5, 50), (20000, 64, 100), (50000, 40, 10), (7777, 17, 128)])
def test_pipelined_and_unpipelined_kernels_agree(gpu, monkeypatch, n, nq, k):
    """Two independent implementations of the scan (software-pipelined with LDS/register lists and the
    global bound vs the plain per-tile kernel kept for A/B runs and as the prefilter path's overflow fallback) must
    agree: same counts, same ids outside fp32 near-ties, scores to 2e-7 (since round 3 both use the same expression
    for the score, so they agree bit for bit; the tolerances stay as the contract)."""
    rng = np.random.default_rng(n + nq + k)
    corpus = unit_rows(rng, n)
    corpus[n // 3] = corpus[5]
    q = rng.standard_normal((nq, 1024)).astype(np.float32)
    mask = rng.random((nq, n)) < 0.7
    index = DenseIndex(1024, capacity=n)
    try:
        index.add(corpus)
        packed = DenseIndex.pack_mask(mask)
        monkeypatch.delenv("CRAG_UNPIPELINED", raising=False)
        a = index.search(q, k, row_mask=packed)
        monkeypatch.setenv("CRAG_UNPIPELINED", "1")
        b = index.search(q, k, row_mask=packed)
        assert_topk_matches(a[0], a[1], a[2], b[0], b[1].astype(np.float64), b[2], tol=2e-7, gap=2e-6)
    finally:
        index.close()


def test_dense_table_loader_accepts_pgvector_text_and_binary(gpu):
    """DenseTable.from_rows: the startup load of `SELECT ..., embedding FROM chunks WHERE embedding IS NOT NULL`
    with embeddings as text literals, pgvector binary and float lists; round-trips bit-exactly."""
    from datetime import datetime
    from uuid import UUID

    from cadence_rag_amd import vector_io
    rng = np.random.default_rng(12)
    n = 300
    vecs = unit_rows(rng, n)
    forms = [lambda v: rt._vector_literal(v.tolist()), lambda v: vector_io.to_binary(v), lambda v: v.tolist()]
    rows = [{"chunk_id": 10 + i, "call_id": UUID(int=1 + i % 3), "speaker": "S", "start_ts_ms": i, "end_ts_ms": i + 1,
             "text": f"t{i}", "call_started_at": datetime(2026, 1, 1 + i % 5), "tech_tokens": ["X-1"] if i % 7 == 0 else [],
             "embedding": forms[i % 3](vecs[i])} for i in range(n)]
    table, tokens = rt.DenseTable.from_rows("chunks", "chunk_id", rows, select=rt.CHUNK_SELECT, dim=1024, batch=128)
    try:
        assert len(table) == n and tokens[7] == ["X-1"] and tokens[1] == []
        back, back_ids = table.index.get_rows(0, n)
        assert np.array_equal(back, vecs) and back_ids.tolist() == list(range(10, 10 + n))  # '.10g' text and big-endian float4 both round-trip float32
        got = rt._fetch_chunks_dense(table, vecs[123], None, None, "exact", 3)
        assert got[0]["chunk_id"] == 133 and got[0]["text"] == "t123"
        lane = table.build_tech_lane(tokens)
        ids, ct = lane.search([["X-1"]], 5)
        assert int(ct[0]) == 5
        with pytest.raises(ValueError, match="ascending"):
            rt.DenseTable.from_rows("chunks", "chunk_id", rows[::-1][:2], select=rt.CHUNK_SELECT, dim=1024)[0].close()
    finally:
        table.close()


def test_property_random_shapes_masks_and_k(gpu):
    """Property test over the whole kernel family (k <= 32 / 64 / 128, <= 32 / > 32 queries, ragged sizes,
    zero rows, sparse and dense masks, small dims): every case must match the fp64 oracle."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    @settings(max_examples=30, deadline=None, derandomize=True,
              suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
    @given(n=st.integers(1, 2500), nq=st.integers(1, 70), k=st.sampled_from([1, 3, 10, 32, 33, 50, 64, 100, 128]),
           dim=st.sampled_from([1024, 1024, 1024, 384, 7]), mask_p=st.sampled_from([None, None, 0.0, 0.02, 0.5, 1.0]),
           zero_rows=st.booleans(), seed=st.integers(0, 2 ** 16))
    def run(n, nq, k, dim, mask_p, zero_rows, seed):
        rng = np.random.default_rng(seed)
        corpus = rng.standard_normal((n, dim)).astype(np.float32)
        if zero_rows and n > 3:
            corpus[rng.integers(0, n, size=2)] = 0.0
        if n > 8:
            corpus[n - 1] = corpus[1]  # an exact duplicate: tie broken by position
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        mask = None if mask_p is None else (rng.random((nq, n)) < mask_p)
        _check(corpus, q, k, mask=mask)

    run()


def test_concurrent_host_threads_and_streams(gpu):
    """crag_index_search is documented thread-safe (FastAPI serves sync endpoints from a thread pool) and
    crag_index_search_async may be used from up to four streams at once: answers must not depend on it."""
    import threading
    import torch
    rng = np.random.default_rng(404)
    n = 30000
    corpus = unit_rows(rng, n)
    qs = [rng.standard_normal((int(nq), 1024)).astype(np.float32) for nq in (1, 7, 33, 64, 20, 3)]
    ks = [10, 50, 10, 10, 100, 128]
    with DenseIndex(1024, capacity=n) as ix:
        ix.add(corpus)
        want = [ix.search(q, k) for q, k in zip(qs, ks)]
        errors = []

        def worker(tid):
            try:
                for rep in range(15):
                    i = (tid + rep) % len(qs)
                    got = ix.search(qs[i], ks[i])
                    for a, b in zip(got, want[i]):
                        assert np.array_equal(a, b, equal_nan=True)
            except Exception as exc:  # noqa: BLE001
                errors.append(repr(exc))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:2]
        # four streams, device buffers, launched back to back without synchronising in between
        dev = torch.device("cuda", 0)
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
        outs = []
        for rep in range(3):
            for si, st in enumerate(streams):
                i = (si + rep) % len(qs)
                dq = torch.from_numpy(qs[i]).to(dev)
                o = (torch.empty(len(qs[i]), ks[i], dtype=torch.int64, device=dev),
                     torch.empty(len(qs[i]), ks[i], dtype=torch.float32, device=dev),
                     torch.empty(len(qs[i]), dtype=torch.int32, device=dev))
                torch.cuda.current_stream().synchronize()  # the H2D copy above ran on the default stream
                ix.search_async(dq, ks[i], *o, stream=st.cuda_stream)
                outs.append((i, o, dq))
        torch.cuda.synchronize()
        for i, o, _ in outs:
            assert np.array_equal(o[0].cpu().numpy(), want[i][0])
            assert np.array_equal(o[1].cpu().numpy(), want[i][1], equal_nan=True)
            assert np.array_equal(o[2].cpu().numpy(), want[i][2])


def test_pipelined_searches_from_one_stream_equal_the_in_order_form(gpu):
    """crag_index_search_pipelined: a run of independent searches issued from ONE stream rotates over the
    streams of the index's own (search i + 1's preparation and scan beside search i's selection); outputs are defined
    behind crag_index_join.  Same bits as crag_index_search_async, for mixed batch sizes and k (fp32 path, prefilter
    path, shared selection), with the inputs produced on the caller's stream right before the call (the fork event
    orders them) and with resident inputs (inputs_ready)."""
    import torch
    rng = np.random.default_rng(911)
    n = 70_000
    corpus = unit_rows(rng, n)
    dev = torch.device("cuda", 0)
    shapes = [(64, 10), (5, 50), (33, 100), (64, 10), (1, 10), (64, 128), (17, 24), (64, 10)]
    with DenseIndex(1024, capacity=n) as ix:
        ix.add(corpus)
        st = torch.cuda.Stream(device=dev)
        host_q = [rng.standard_normal((nq, 1024)).astype(np.float32) for nq, _ in shapes]
        want = [ix.search(q, k) for q, (_, k) in zip(host_q, shapes)]
        for ready in (False, True):
            outs = []
            with torch.cuda.stream(st):
                dqs = []
                staging = [torch.from_numpy(q).to(dev) for q in host_q]
                st.synchronize()
                for src in staging:
                    # not ready: the query block is PRODUCED on the stream right before the searches (a copy kernel behind
                    # a few milliseconds of other work), so only the fork event keeps the scan from reading it early
                    if not ready:
                        busy = torch.randn(4096, 4096, device=dev)
                        busy = busy @ busy
                    dqs.append(src.clone())
                if ready:
                    st.synchronize()     # resident inputs: what inputs_ready promises
                for rep in range(3):
                    for dq, (nq, k) in zip(dqs, shapes):
                        o = (torch.empty(nq, k, dtype=torch.int64, device=dev), torch.empty(nq, k, dtype=torch.float32, device=dev),
                             torch.empty(nq, dtype=torch.int32, device=dev))
                        ix.search_pipelined(dq, k, *o, stream=st.cuda_stream, inputs_ready=ready)
                        outs.append(o)
                ix.join(st.cuda_stream)
                host = [tuple(t.cpu().numpy() for t in o) for o in outs]   # (copies enqueued on `st`, behind the join)
            for i, got in enumerate(host):
                w = want[i % len(shapes)]
                assert np.array_equal(got[0], w[0]) and np.array_equal(got[2], w[2])
                assert np.array_equal(got[1], w[1], equal_nan=True)
        # the in-order form still works on the same index afterwards, and a join with nothing pending is a no-op
        ix.join(0)
        again = ix.search(host_q[0], shapes[0][1])
        for a, b in zip(again, want[0]):
            assert np.array_equal(a, b, equal_nan=True)


def test_overlapped_scans_keep_their_bounds_and_stay_out_of_the_fallback(gpu):
    """Scans that share the GPU get their workgroups a few at a time, and a query's bound is derived by four particular
    workgroups (its delegates): while none of them runs, the others must derive it themselves (the emergency derivation
    inside a workgroup) -- otherwise they pass every row and the search ends in the overflow fallback (what the first
    version of the delegated kernel did with two 1M-row scans started together: 1.9 ms per search instead of 0.33).
    Twelve searches over 500 000 rows on three internal streams: every one takes the selection path (the statistics
    count only those), same bits as in order."""
    import torch
    dev = torch.device("cuda", 0)
    n = 500_000
    g = torch.Generator(device=dev).manual_seed(99)
    corpus = torch.randn(n, 1024, generator=g, device=dev, dtype=torch.float32)
    q = torch.randn(64, 1024, generator=g, device=dev, dtype=torch.float32)
    with DenseIndex(1024, capacity=n) as ix:
        ix.add(corpus)
        del corpus
        want = tuple(torch.empty(s, dtype=d, device=dev) for s, d in (((64, 10), torch.int64), ((64, 10), torch.float32), ((64,), torch.int32)))
        st = torch.cuda.current_stream().cuda_stream
        ix.search_async(q, 10, *want, stream=st)
        torch.cuda.synchronize()
        ix.prefilter_stats()
        outs = [tuple(torch.empty_like(t) for t in want) for _ in range(12)]
        for o in outs:
            ix.search_pipelined(q, 10, *o, stream=st, inputs_ready=True)
        ix.join(st)
        torch.cuda.synchronize()
        stats = ix.prefilter_stats()
        assert "prefilter" in ix.last_scan_kernel()
        assert stats["searches"] == 12, stats          # none of them overflowed into the fallback
        assert stats["candidates"] / 12 / 64 < 3000, stats  # ... nor came close to the 8192-entry lists (in order: ~40 per
                                                            # query; overlapped, with the emergency derivation: ~600)
        for o in outs:
            for a, b in zip(o, want):
                assert torch.equal(a, b)
