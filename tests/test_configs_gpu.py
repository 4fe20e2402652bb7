"""GPU tests at the sizes BASELINE.json's configs name (full-size 1M x 1024 scans, the 4B encoder widths,
the embed_backfill batch shape) and for the boundary rules the C ABI documents (ascending ids, stream
take-over, pgvector-verbatim fp32 arithmetic).  Everything goes through the C ABI / the product package;
the oracle is the checker only."""
import threading

import numpy as np
import pytest

import oracle
from cadence_rag_amd.dense_index import DenseIndex
from tests.helpers import assert_topk_matches, unit_rows

pytestmark = pytest.mark.gpu
TOL = 1e-4  # BASELINE.json: cosine scores within 1e-4 (fp32)


# ------------------------------------------------------------------------------------------------------
# configs[2] / configs[4]: 1M x 1024 (4.1 GB) on one GPU — size-independent properties + sampled oracle
# ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def million(gpu):
    """1M x 1024 unit rows generated on the device (seed 1234) with 64 planted near-duplicates, a block of
    exact duplicates and a few ineligible rows; the host copy feeds the sampled oracle."""
    import torch
    dev = torch.device("cuda", 0)
    n = 1_000_000
    g = torch.Generator(device=dev).manual_seed(1234)
    corpus = torch.empty(n, 1024, dtype=torch.float32, device=dev)
    for lo in range(0, n, 125_000):
        blk = torch.randn(125_000, 1024, generator=g, device=dev)
        corpus[lo:lo + 125_000] = blk / blk.norm(dim=1, keepdim=True)
    corpus[777_000:777_040] = corpus[5]          # 40 exact duplicates of row 5 far away from it
    corpus[123_456] = 0.0                        # zero row: never eligible
    corpus[654_321, 17] = float("nan")           # NaN row: never eligible
    rng = np.random.default_rng(4321)
    planted = np.sort(rng.choice(n, size=64, replace=False))
    planted = planted[~np.isin(planted, [5, 123_456, 654_321]) & ~((planted >= 777_000) & (planted < 777_040))]
    noise = torch.from_numpy(rng.standard_normal((len(planted), 1024)).astype(np.float32)).to(dev)
    queries = corpus[torch.from_numpy(planted).to(dev)] + 0.05 * noise / 32.0
    queries = torch.cat([queries, corpus[5:6] * 2.5,
                         torch.from_numpy(rng.standard_normal((64 - len(planted) - 1 + 16, 1024)).astype(np.float32)).to(dev)])
    ix = DenseIndex(1024, capacity=n)
    ix.add(corpus)
    host = corpus.cpu().numpy()
    q_host = queries.cpu().numpy()
    del corpus
    torch.cuda.empty_cache()
    yield ix, host, q_host, planted
    ix.close()


def _sampled_oracle(host, q, k, rows):
    """fp64 oracle for the query rows `rows` only (a full 1M oracle pass per query is ~1 s of host time)."""
    oracle.set_threads(16)
    return oracle.exact_topk(q[rows], host, k, mode=oracle.F64, fast=True)


@pytest.mark.parametrize("nq,k", [(32, 10), (64, 10), (64, 100)])
def test_one_million_rows_properties_and_sampled_oracle(million, nq, k):
    ix, host, q_all, planted = million
    q = q_all[:nq]
    ids, scores, counts = ix.search(q, k)
    ids2, scores2, counts2 = ix.search(q, k)                   # the reversed pass over the same corpus
    assert np.array_equal(ids, ids2) and np.array_equal(scores, scores2) and np.array_equal(counts, counts2)
    assert np.all(counts == k) and np.all(np.diff(scores, axis=1) <= 0)
    npl = min(nq, len(planted))
    assert np.array_equal(ids[:npl, 0], planted[:npl])         # every planted neighbour is found first
    assert not np.isin(ids, [123_456, 654_321]).any()          # zero / NaN rows never come back
    for row in range(nq):                                      # ids are distinct inside a result
        assert len(set(ids[row].tolist())) == k
    sample = sorted({0, 1, nq // 4, nq // 2, (3 * nq) // 4, nq - 2, nq - 1})   # (round 2 checked three per shape)
    want = _sampled_oracle(host, q, k, sample)
    assert_topk_matches(ids[sample], scores[sample], counts[sample], *want, tol=TOL)
    recall = np.mean([len(set(ids[r][:10]) & set(want[0][i][:10])) / 10 for i, r in enumerate(sample)])
    assert recall == 1.0
    # one query at a time == the batched answer (the 1-query, 32-query and 64-query kernels agree)
    mid = nq // 2
    one = ix.search(q[mid][None], k)
    assert np.array_equal(one[0][0], ids[mid])
    assert np.max(np.abs(one[1][0] - scores[mid])) <= 2e-6


def test_one_million_rows_exact_duplicates_order_by_id(million):
    ix, host, q_all, planted = million
    qi = len(planted)                                          # the query that is 2.5 x row 5
    ids, scores, counts = ix.search(q_all[qi][None], 50)
    assert ids[0, 0] == 5 and list(ids[0, 1:41]) == list(range(777_000, 777_040))
    assert np.all(scores[0, :41] == scores[0, 0])              # bit-identical scores for identical rows
    # a shared row mask that removes the low half of the duplicates
    mask = np.ones(1_000_000, dtype=bool)
    mask[777_000:777_020] = False
    mask[5] = False
    ids, _, _ = ix.search(q_all[qi][None], 50, row_mask=DenseIndex.pack_mask(mask))
    assert list(ids[0, :20]) == list(range(777_020, 777_040))
    assert ix.count_eligible() == 1_000_000 - 2


def test_hybrid_retrieve_at_full_size_one_million_chunks_batch_64(million):
    """BASELINE configs[4] at its full size: 1M chunks, batch 64, dense top-100 + exact-token lane top-50 + given
    BM25 ranks -> RRF, all on the GPU (fusion.HybridSearcher).  Checked: the dense leg against a sampled fp64 oracle;
    the token lane against a vectorised host evaluation of `tech_tokens && :tokens ORDER BY call_started_at DESC,
    id ASC LIMIT 50` (retrieve.py:183-242) for every query; the fused order against the host mirror of _rrf_merge
    (retrieve.py:245-260, itself pinned by goldens from the reference) fed with the GPU lanes, for every query; and
    the size-independent properties of RRF (score = sum of 1/(60 + rank) over the lanes that hit, descending,
    distinct ids, lane bits)."""
    import torch
    from cadence_rag_amd import retrieve as rt
    from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex
    ix, host, q_all, planted = million
    n, nq = 1_000_000, 64
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    vocab = np.array([f"TOK-{i}" for i in range(3000)])
    n_tok = rng.integers(0, 4, size=n)
    tok_idx = rng.integers(0, 3000, size=int(n_tok.sum()))
    row_of_tok = np.repeat(np.arange(n), n_tok)
    flat = vocab[tok_idx].tolist()
    row_tokens, o = [], 0
    for c in n_tok.tolist():
        row_tokens.append(flat[o:o + c])
        o += c
    started = np.datetime64("2025-01-01", "us") + rng.integers(0, 500, size=n).astype("timedelta64[D]")
    tech = TechTokenIndex(row_tokens, np.arange(n), started, dev, verify=False)
    q_tok_idx = [rng.choice(3000, size=int(rng.integers(0, 4)), replace=False) for _ in range(nq)]
    qtoks = [vocab[t].tolist() for t in q_tok_idx]
    bm25 = np.stack([rng.choice(n, size=50, replace=False) for _ in range(nq)])
    bm25_ct = rng.integers(0, 51, size=nq).astype(np.int32)
    q = q_all[:nq]
    hs = HybridSearcher(ix, tech, dense_k=100, tech_k=50)
    out = hs.search(torch.from_numpy(q).to(dev), qtoks, (torch.from_numpy(bm25).to(dev), torch.from_numpy(bm25_ct).to(dev)))
    torch.cuda.synchronize()
    ids, cts = out["ids"].cpu().numpy(), out["counts"].cpu().numpy()
    scs, lanes = out["scores"].cpu().numpy(), out["lanes"].cpu().numpy()
    d_ids, d_sc, d_ct = (out[k].cpu().numpy() for k in ("dense_ids", "dense_scores", "dense_counts"))
    # dense leg: top-100 vs the sampled fp64 oracle
    sample = [0, 31, 63]
    want = _sampled_oracle(host, q, 100, sample)
    assert_topk_matches(d_ids[sample], d_sc[sample], d_ct[sample], *want, tol=TOL)
    # token lane, every query: rows holding any query token, most recent call first, then id
    t_ids, t_ct = (t.cpu().numpy() for t in tech.search(qtoks, 50))
    order_key = np.lexsort((np.arange(n), -started.astype(np.int64)))        # rank -> row
    rank_of = np.empty(n, dtype=np.int64)
    rank_of[order_key] = np.arange(n)
    for qi in range(nq):
        hit_rows = np.unique(row_of_tok[np.isin(tok_idx, q_tok_idx[qi])]) if len(q_tok_idx[qi]) else np.empty(0, np.int64)
        want_rows = hit_rows[np.argsort(rank_of[hit_rows])][:50]
        assert t_ct[qi] == len(want_rows)
        assert t_ids[qi, :t_ct[qi]].tolist() == want_rows.tolist()
    # fusion, every query: the GPU lanes through the host mirror of _rrf_merge
    bits = {"bm25": 1, "tech_tokens": 2, "dense": 4}
    for qi in range(nq):
        lanes_in = {"bm25": [{"id": int(v)} for v in bm25[qi, :bm25_ct[qi]]],
                    "tech_tokens": [{"id": int(v)} for v in t_ids[qi, :t_ct[qi]]],
                    "dense": [{"id": int(v)} for v in d_ids[qi, :d_ct[qi]]]}
        ref = rt._rrf_merge(lanes_in, "id")
        c = int(cts[qi])
        assert c == len(ref) and len(set(ids[qi, :c].tolist())) == c
        assert ids[qi, :c].tolist() == [row["id"] for row, _, _ in ref]
        assert np.array_equal(scs[qi, :c], np.array([sc for _, _, sc in ref]))
        assert lanes[qi, :c].tolist() == [sum(bits[l] for l in ls) for _, ls, _ in ref]
        assert np.all(np.diff(scs[qi, :c]) <= 0)
        # a row that only the dense lane found at rank r scores exactly 1/(60 + r)
        only_dense = [i for i in range(c) if lanes[qi, i] == 4]
        for i in only_dense[:5]:
            r = int(np.nonzero(d_ids[qi] == ids[qi, i])[0][0]) + 1
            assert scs[qi, i] == 1.0 / (60 + r)


# ------------------------------------------------------------------------------------------------------
# boundary rules of include/crag_dense.h
# ------------------------------------------------------------------------------------------------------
def test_ids_must_ascend_with_the_row_position(gpu):
    """ties by ascending id.  The scan breaks ties on the row position, so crag_index_add
    refuses ids that do not grow with it (host and device pointers), leaving the index unchanged."""
    import torch
    rng = np.random.default_rng(2)
    rows = unit_rows(rng, 64)
    with DenseIndex(1024, capacity=1000) as ix:
        ix.add(rows[:10], ids=np.arange(100, 110))
        for bad in (np.arange(105, 115), np.array([200, 199] + list(range(300, 308))),
                    np.array([300] * 10), np.arange(109, 119)):
            with pytest.raises(Exception, match="strictly ascending"):
                ix.add(rows[10:20], ids=bad.astype(np.int64))
            with pytest.raises(Exception, match="strictly ascending"):
                ix.add(torch.from_numpy(rows[10:20]).cuda(), ids=torch.from_numpy(bad.astype(np.int64)).cuda())
            assert len(ix) == 10
        with pytest.raises(Exception, match="implicit ids"):
            ix.add(rows[10:20])                                 # implicit ids would start at 10 <= 109
        ix.add(rows[10:20], ids=np.arange(110, 120))
        ix.add(torch.from_numpy(rows[20:30]).cuda(), ids=torch.arange(500, 510).cuda())
        assert len(ix) == 30
        got = ix.search(rows[25][None], 3)
        assert got[0][0, 0] == 505
        _, stored = ix.get_rows(0, 30)
        assert stored.tolist() == list(range(100, 120)) + list(range(500, 510))


def test_dense_table_insert_out_of_order_rebuilds_in_id_order(gpu):
    """A row embedded late (id below the stored maximum) goes in through DenseTable.insert; ties then
    still resolve to the lower id, filters and fetches see the merged table, and the index grows."""
    from uuid import UUID
    from cadence_rag_amd import retrieve as rt
    rng = np.random.default_rng(3)
    vecs = unit_rows(rng, 300)
    vecs[250] = vecs[40]                                        # duplicate vectors: ids 1040 and 90
    table = rt.DenseTable("chunks", "chunk_id", dim=1024, capacity=200)
    try:
        def cols(ids):
            return {"chunk_id": list(ids), "call_id": [UUID(int=1 + (i % 3)) for i in ids],
                    "speaker": ["S"] * len(ids), "start_ts_ms": [0] * len(ids), "end_ts_ms": [1] * len(ids),
                    "text": [f"row {i}" for i in ids]}
        table.add(vecs[:200], cols(range(1000, 1200)))
        assert table.index.capacity == 200
        sink = table.sink(lambda ids: cols(ids))
        sink.add(vecs[200:260].tolist(), ids=list(range(50, 110)))   # late rows: ids BELOW the stored ones
        sink.add(vecs[260:300].tolist(), ids=list(range(1200, 1240)))  # and a plain append past the capacity
        assert len(table) == 300 and table.index.capacity >= 300
        assert table.columns["chunk_id"] == list(range(50, 110)) + list(range(1000, 1240))
        rows = rt._fetch_chunks_dense(table, vecs[40] * 2.0, None, None, "exact", 5)
        assert [r["chunk_id"] for r in rows[:2]] == [100, 1040] and rows[0]["text"] == "row 100"
        assert rows[0]["score"] == rows[1]["score"]
        f = rt.RetrieveFilters(call_ids=[UUID(int=2)])
        scoped = rt._fetch_chunks_dense(table, vecs[40], f, f.call_ids, "exact", 128)
        assert scoped and all(r["call_id"] == UUID(int=2) for r in scoped)
        assert rt._estimate_dense_candidates(table, "chunks", f, f.call_ids) == sum(
            1 for i in table.columns["chunk_id"] if 1 + (i % 3) == 2)
        order = np.argsort(np.r_[np.arange(1000, 1200), np.arange(50, 110), np.arange(1200, 1240)], kind="stable")
        want = oracle.exact_topk(vecs[7][None], vecs[order], 10, ids=np.asarray(table.columns["chunk_id"]),
                                 mode=oracle.F64)
        got = table.index.search(vecs[7][None], 10)
        assert_topk_matches(*got, *want, tol=TOL)
        with pytest.raises(ValueError, match="duplicate"):
            sink.add(vecs[:1].tolist(), ids=[1005])
    finally:
        table.close()


def test_device_resident_late_rows_and_the_token_lane_follow_the_table(gpu):
    """(1) a CUDA tensor of late-embedded rows (ids below the stored maximum) goes through
    DenseTable.sink / insert without visiting the host (DeviceSinkStore hands exactly that); (2) the exact-token lane
    is built over row POSITIONS: once rows were appended or moved it is stale, and GpuRetrieveBackend rebuilds it from
    the tokens the table tracks, so filters and token hits keep meaning the table's rows."""
    import torch
    from datetime import datetime
    from uuid import UUID
    from cadence_rag_amd import retrieve as rt
    rng = np.random.default_rng(12)
    vecs = unit_rows(rng, 120)
    dev = torch.device("cuda", 0)
    table = rt.DenseTable("chunks", "chunk_id", dim=1024, capacity=64)
    try:
        def cols(ids):
            ids = list(ids)
            return {"chunk_id": ids, "call_id": [UUID(int=1 + (i % 2)) for i in ids], "speaker": ["S"] * len(ids),
                    "start_ts_ms": [0] * len(ids), "end_ts_ms": [1] * len(ids), "text": [f"row {i}" for i in ids],
                    "call_started_at": [datetime(2024, 1, 1 + (i % 20)) for i in ids],
                    "tech_tokens": [["ECONNRESET"] if i % 10 == 0 else [f"TOK-{i}"] for i in ids]}
        first = cols(range(1000, 1060))
        started, toks = first.pop("call_started_at"), first.pop("tech_tokens")
        table.add(vecs[:60], first, call_started_at=started)
        lane = table.build_tech_lane(toks)
        backend = rt.GpuRetrieveBackend(table, rt.DenseTable("artifact_chunks", "artifact_chunk_id", dim=1024, capacity=1),
                                        tech_chunks=lane)
        before = [r["chunk_id"] for r in backend.fetch_chunks_tech(["ECONNRESET"], None, None, 50)]
        assert sorted(before) == [i for i in range(1000, 1060) if i % 10 == 0]
        gen = table.generation
        sink = table.sink(lambda ids: cols(ids))
        sink.add(torch.from_numpy(vecs[60:100]).to(dev), ids=list(range(500, 540)))      # late rows, CUDA tensor
        sink.add(torch.from_numpy(vecs[100:120]).to(dev), ids=list(range(1060, 1080)))   # plain append, CUDA tensor
        assert table.generation > gen and len(table) == 120
        assert table.columns["chunk_id"] == list(range(500, 540)) + list(range(1000, 1080))
        stored, ids = table.index.get_rows(0, 120)
        assert np.array_equal(stored, np.concatenate([vecs[60:100], vecs[:60], vecs[100:120]]))
        assert ids.tolist() == table.columns["chunk_id"]
        # the stale lane is rebuilt: new rows are found, the order is (call_started_at DESC, id ASC), and a call_id
        # filter (packed by CURRENT positions) selects exactly that call's rows
        hits = backend.fetch_chunks_tech(["ECONNRESET"], None, None, 50)
        want = sorted((i for i in table.columns["chunk_id"] if i % 10 == 0),
                      key=lambda i: (-(1 + i % 20), i))
        assert [r["chunk_id"] for r in hits] == want
        assert backend._tech["chunks"].table_generation == table.generation
        f = rt.RetrieveFilters(call_ids=[UUID(int=1)])
        scoped = backend.fetch_chunks_tech(["ECONNRESET", "TOK-501", "TOK-1077"], f, f.call_ids, 50)
        assert scoped and all(r["call_id"] == UUID(int=1) for r in scoped)
        assert {r["chunk_id"] for r in scoped} == {i for i in want if i % 2 == 0}
        # a table that does not track tokens refuses to answer from a stale lane
        table.tech_tokens = None
        table.add(vecs[:1], {k: v for k, v in cols([2000]).items() if k not in ("call_started_at", "tech_tokens")})
        with pytest.raises(RuntimeError, match="stale"):
            backend.fetch_chunks_tech(["ECONNRESET"], None, None, 5)
    finally:
        table.close()


def test_more_streams_than_workspaces_take_over_behind_events(gpu):
    """The index keeps 8 search workspaces (the host-buffer search above took the first); the 8th .. 11th stream takes
    over the least recently used one -- the first time behind a device synchronisation (no event was recorded while
    workspaces were free), from then on behind the completion event of its last search.  Eleven streams cycling three
    times, nothing synchronised in between: every answer equals the single-stream answer."""
    import torch
    rng = np.random.default_rng(77)
    n = 40_000
    corpus = unit_rows(rng, n)
    shapes = [(64, 10), (33, 50), (5, 10), (64, 100), (1, 128), (32, 10), (40, 64)]
    qs = [rng.standard_normal((nq, 1024)).astype(np.float32) for nq, _ in shapes]
    dev = torch.device("cuda", 0)
    with DenseIndex(1024, capacity=n) as ix:
        ix.add(corpus)
        want = [ix.search(q, k) for q, (_, k) in zip(qs, shapes)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(11)]
        dqs = [torch.from_numpy(q).to(dev) for q in qs]
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):
            for si, st in enumerate(streams):
                i = (si + 2 * rep) % len(qs)
                nq, k = shapes[i]
                o = (torch.empty(nq, k, dtype=torch.int64, device=dev), torch.empty(nq, k, dtype=torch.float32, device=dev),
                     torch.empty(nq, dtype=torch.int32, device=dev))
                ix.search_async(dqs[i], k, *o, stream=st.cuda_stream)
                outs.append((i, o))
        torch.cuda.synchronize()
        for i, o in outs:
            assert np.array_equal(o[0].cpu().numpy(), want[i][0])
            assert np.array_equal(o[1].cpu().numpy(), want[i][1], equal_nan=True)
            assert np.array_equal(o[2].cpu().numpy(), want[i][2])


def test_add_while_searches_are_in_flight(gpu):
    """include/crag_dense.h: appending is safe while searches enqueued earlier are still running, because a
    scan never reads past the size it was launched with."""
    import torch
    rng = np.random.default_rng(78)
    base, extra = unit_rows(rng, 60_000), unit_rows(rng, 2_000)
    q = rng.standard_normal((64, 1024)).astype(np.float32)
    extra[:64] = q / np.linalg.norm(q, axis=1, keepdims=True)     # the appended rows would win every query
    dev = torch.device("cuda", 0)
    with DenseIndex(1024, capacity=62_000) as ix:
        ix.add(base)
        want = ix.search(q, 10)
        st = torch.cuda.Stream(device=dev)
        dq = torch.from_numpy(q).to(dev)
        d_extra = torch.from_numpy(extra).to(dev)
        torch.cuda.synchronize()
        outs = []
        for _ in range(6):
            o = (torch.empty(64, 10, dtype=torch.int64, device=dev), torch.empty(64, 10, dtype=torch.float32, device=dev),
                 torch.empty(64, dtype=torch.int32, device=dev))
            ix.search_async(dq, 10, *o, stream=st.cuda_stream)
            outs.append(o)
        ix.add(d_extra)                                           # null stream, while the six scans are queued
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o[0].cpu().numpy(), want[0]) and np.array_equal(o[1].cpu().numpy(), want[1])
        after = ix.search(q, 10)
        assert np.array_equal(after[0][:, 0], np.arange(60_000, 60_064))


def test_against_pgvector_verbatim_fp32_arithmetic(gpu):
    """oracle F32SEQ = pgvector 0.8.1's own arithmetic (sequential fp32 accumulators).  The HIP path sums in
    another fixed order, so scores differ in the last bits; reported: how many top-10 lists keep pgvector's
    order, and that every difference is a swap of rows pgvector itself scores within 2e-6 of each other."""
    rng = np.random.default_rng(2026)
    n, nq, k = 50_000, 64, 10
    corpus, q = unit_rows(rng, n), rng.standard_normal((nq, 1024)).astype(np.float32)
    with DenseIndex(1024, capacity=n) as ix:
        ix.add(corpus)
        ids, scores, counts = ix.search(q, k)
    want = oracle.exact_topk(q, corpus, k, mode=oracle.F32SEQ)
    same = sum(np.array_equal(ids[i], want[0][i]) for i in range(nq))
    print(f"\npgvector-verbatim fp32 order reproduced for {same}/{nq} top-{k} lists; "
          f"max |dscore| = {np.max(np.abs(scores - want[1])):.2e}")
    assert_topk_matches(ids, scores, counts, *want, tol=TOL)      # only near-tie permutations differ
    assert same >= nq - 3
    assert np.max(np.abs(scores - want[1])) <= 5e-6


def test_tech_lane_with_more_than_32_query_tokens(gpu):
    """`tech_tokens && :tokens` has no bound on the query's token count (retrieve.py:183-242); the kernel takes
    32 per launch, longer lists run in passes merged in the lane's order."""
    import torch
    from cadence_rag_amd.fusion import TechTokenIndex
    rng = np.random.default_rng(9)
    n = 5000
    vocab = [f"TOK-{i}" for i in range(400)]
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
    ids = np.arange(n, dtype=np.int64) * 3 + 7
    started = np.datetime64("2026-03-01", "us") + rng.integers(0, 40, size=n).astype("timedelta64[D]")
    queries = [list(rng.choice(vocab, size=m, replace=False)) for m in (70, 33, 5, 0, 100)]
    queries[1] = queries[1] + queries[1][:4]                     # repeated tokens do not count twice
    tech = TechTokenIndex(row_tokens, ids, started, torch.device("cuda", 0))
    got_ids, got_ct = tech.search(queries, 25)
    order = np.lexsort((ids, -started.astype(np.int64)))
    for qi, toks in enumerate(queries):
        want = [int(ids[p]) for p in order if set(row_tokens[p]) & set(toks)][:25]
        assert got_ids[qi, :int(got_ct[qi])].tolist() == want


def test_tech_lane_from_host_hashes_in_one_call(gpu):
    """verify=False (the stream-ordered path HybridSearcher takes): crag_tech_lane_host packs the hashes in the library's
    own upload slots -- repeated tokens inside a query dropped there, 32 distinct tokens + repeats still one pass, a
    query with more distinct tokens falls back to passes, empty lists, a row mask, ten calls in a row over the ring of
    four slots with different queries, borrowed and fresh outputs."""
    import torch
    from cadence_rag_amd.fusion import TechTokenIndex
    rng = np.random.default_rng(31)
    n = 6000
    vocab = [f"T-{i}" for i in range(300)]
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
    ids = np.arange(n, dtype=np.int64) * 2 + 11
    started = np.datetime64("2026-03-01", "us") + rng.integers(0, 30, size=n).astype("timedelta64[D]")
    dev = torch.device("cuda", 0)
    tech = TechTokenIndex(row_tokens, ids, started, dev, verify=False)
    order = np.lexsort((ids, -started.astype(np.int64)))
    elig = rng.random(n) < 0.6
    mask = torch.from_numpy(DenseIndex.pack_mask(elig)).to(dev)

    def want(toks, k, use_mask):
        return [int(ids[p]) for p in order if set(row_tokens[p]) & set(toks) and (elig[p] or not use_mask)][:k]

    def check(queries, k, use_mask, borrow):
        got_ids, got_ct = tech.search(queries, k, row_mask=mask if use_mask else None, verify=False, borrow=borrow)
        torch.cuda.synchronize()
        gi, gc = got_ids.cpu().numpy(), got_ct.cpu().numpy()
        assert gi.shape == (len(queries), k)
        for qi, toks in enumerate(queries):
            w = want(toks, k, use_mask)
            assert int(gc[qi]) == len(w) and gi[qi, :len(w)].tolist() == w and np.all(gi[qi, len(w):] == -1), qi

    full = list(rng.choice(vocab, size=32, replace=False))
    one_pass = [full + full[:9], [], ["T-5", "T-5", "T-5"], ["not-a-token"], list(rng.choice(vocab, size=3, replace=False))]
    for borrow in (False, True):
        for use_mask in (False, True):
            check(one_pass, 20, use_mask, borrow)
    check([list(rng.choice(vocab, size=33, replace=False)), ["T-1"]], 20, False, False)     # 33 distinct: passes
    for i in range(10):                                           # the ring of four slots, a different batch every call
        check([list(rng.choice(vocab, size=int(m), replace=True)) for m in rng.integers(0, 12, size=1 + i % 7)], 7, i % 2 == 1,
              True)
    a = tech.search(one_pass, 20, verify=False)                   # fresh outputs survive later calls
    for _ in range(6):
        tech.search([["T-9"]], 20, verify=False, borrow=True)
    torch.cuda.synchronize()
    assert a[0][0, :int(a[1][0])].tolist() == want(one_pass[0], 20, False)
    tech.close()


def test_tech_lane_hash_collisions_are_repaired_from_the_strings(gpu, monkeypatch):
    """The kernel matches 64-bit token hashes; TechTokenIndex (verify=True, the default) checks every returned row
    against the token STRINGS and re-evaluates a query on a mismatch, so the lane has the SQL `&&` semantics
    exactly.  Forced here with a 2-bit hash."""
    import torch
    from cadence_rag_amd import fusion
    monkeypatch.setattr(fusion, "token_hash", lambda t: 1 + (sum(t.encode()) % 4))
    rng = np.random.default_rng(13)
    n = 3000
    vocab = [f"W{i}" for i in range(50)]
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
    ids = np.arange(n, dtype=np.int64) + 100
    started = np.datetime64("2026-05-01", "us") + rng.integers(0, 20, size=n).astype("timedelta64[D]")
    queries = [["W3"], ["W7", "W8"], [], ["nope"]]
    order = np.lexsort((ids, -started.astype(np.int64)))
    want = [[int(ids[p]) for p in order if set(row_tokens[p]) & set(q)][:20] for q in queries]
    dev = torch.device("cuda", 0)
    exact = fusion.TechTokenIndex(row_tokens, ids, started, dev)
    got_ids, got_ct = exact.search(queries, 20)
    assert [got_ids[i, :int(got_ct[i])].tolist() for i in range(4)] == want
    loose = fusion.TechTokenIndex(row_tokens, ids, started, dev, verify=False)
    l_ids, l_ct = loose.search(queries, 20)
    assert [l_ids[i, :int(l_ct[i])].tolist() for i in range(4)] != want   # the weak hash really collides


def test_hybrid_searcher_on_two_side_streams(gpu):
    """HybridSearcher / TechTokenIndex keep their scratch per stream and allocate on the caller's stream: two
    side streams sharing one searcher, launched back to back, give the default-stream answers."""
    import torch
    from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex
    rng = np.random.default_rng(10)
    n, nq = 20_000, 16
    corpus = unit_rows(rng, n)
    vocab = [f"T{i}" for i in range(60)]
    row_tokens = [list(rng.choice(vocab, size=rng.integers(0, 3), replace=False)) for _ in range(n)]
    ids = np.arange(n, dtype=np.int64)
    started = np.datetime64("2026-01-01", "us") + rng.integers(0, 9, size=n).astype("timedelta64[D]")
    dev = torch.device("cuda", 0)
    batches = []
    for b in range(2):
        qv = torch.from_numpy(rng.standard_normal((nq, 1024)).astype(np.float32)).to(dev)
        qt = [list(rng.choice(vocab, size=rng.integers(0, 4), replace=False)) for _ in range(nq)]
        batches.append((qv, qt))
    with DenseIndex(1024, capacity=n) as index:
        index.add(corpus, ids=ids)
        hs = HybridSearcher(index, TechTokenIndex(row_tokens, ids, started, dev), dense_k=30, tech_k=20)
        want = []
        for qv, qt in batches:
            out = hs.search(qv, qt)
            torch.cuda.synchronize()
            want.append({key: v.clone() for key, v in out.items()})
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        got = [None, None]
        for rep in range(3):
            for b, st in enumerate(streams):
                got[b] = hs.search(*batches[b], stream=st.cuda_stream)
        torch.cuda.synchronize()
        for b in range(2):
            for key in ("ids", "counts", "dense_ids", "dense_counts"):
                assert torch.equal(got[b][key], want[b][key]), key


def test_sharded_search_over_rccl_world_of_one(gpu):
    """RCCL on hardware: a 1-rank `nccl` group, the packed ResultRecord path forced (one all_gather_into_tensor
    + crag_merge_topk_packed), against the plain search of the same index."""
    import os
    import torch
    import torch.distributed as dist
    from cadence_rag_amd.sharded import ShardedSearch
    rng = np.random.default_rng(55)
    corpus, q = unit_rows(rng, 30_000), rng.standard_normal((64, 1024)).astype(np.float32)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        with DenseIndex(1024, capacity=30_000) as ix:
            ix.add(corpus, ids=np.arange(30_000) + 10**10)
            want = ix.search(q, 10)
            sh = ShardedSearch(ix)
            dq = torch.from_numpy(q).to(dev)
            for _ in range(3):
                ids, sc, ct = sh.search(dq, 10, force_exchange=True)
            torch.cuda.synchronize()
            assert np.array_equal(ids.cpu().numpy(), want[0]) and np.array_equal(sc.cpu().numpy(), want[1])
            assert np.array_equal(ct.cpu().numpy(), want[2])
    finally:
        dist.destroy_process_group()
