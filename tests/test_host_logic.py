"""Host-side mirror of the reference's dense-path Python (embeddings / embedding_pipeline /
retrieve helpers) against values captured from the reference itself
(tests/golden/reference_host_logic.json, made by tests/golden/make_goldens.py) and against the
behaviours the reference's own unit tests assert (tests/unit/test_embeddings_client.py,
test_embedding_pipeline.py, test_retrieve_planner.py)."""
import json
import struct
from datetime import datetime, timezone
from pathlib import Path
from uuid import UUID, uuid4

import pytest

from cadence_rag_amd import embedding_pipeline, embeddings
from cadence_rag_amd import retrieve as rt
from cadence_rag_amd.config import Settings, settings
from cadence_rag_amd.embeddings import EmbeddingClientError, EmbeddingResult

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "reference_host_logic.json").read_text())


# ---- settings ---------------------------------------------------------------------------------
def test_settings_defaults_match_reference_config():
    s = Settings()
    assert (s.embeddings_base_url, s.embeddings_model_id, s.embeddings_dim) == ("", "Qwen/Qwen3-Embedding-4B", 1024)
    assert (s.embeddings_timeout_s, s.embeddings_batch_size) == (180.0, 32)
    assert (s.embeddings_exact_scan_threshold, s.embeddings_hnsw_ef_search) == (2000, 80)
    assert s.ingest_auto_embed_on_success is True and s.ingest_auto_embed_fail_on_error is False


def test_settings_env_is_case_insensitive(monkeypatch):
    monkeypatch.setenv("embeddings_dim", "256")
    monkeypatch.setenv("EMBEDDINGS_BASE_URL", "native")
    s = Settings.from_env()
    assert s.embeddings_dim == 256 and s.embeddings_base_url == "native"


# ---- S1 vector literal --------------------------------------------------------------------------
def test_vector_literal_matches_reference_and_roundtrips_f32():
    for case in GOLD["vector_literal"]:
        vals = [struct.unpack("<f", bytes.fromhex(h))[0] for h in case["values_f32_hex"]]
        lit = rt._vector_literal(vals)
        assert lit == case["literal"]
        assert embedding_pipeline._vector_literal(vals) == case["literal"]
        back = rt._parse_vector(lit)
        assert [struct.pack("<f", float(v)).hex() for v in back] == [h if h != "00000080" else "00000080"
                                                                    for h in case["values_f32_hex"]]


# ---- S2 planner -----------------------------------------------------------------------------------
def _scope(label):
    call = UUID("11111111-2222-3333-4444-555555555555")
    now = datetime(2026, 1, 2, 3, 4, 5, tzinfo=timezone.utc)
    return {
        "none": (None, None),
        "empty_filters": (rt.RetrieveFilters(), None),
        "call_ids": (rt.RetrieveFilters(call_ids=[call]), [call]),
        "resolved_empty": (rt.RetrieveFilters(external_id="x"), []),
        "dates": (rt.RetrieveFilters(date_from=now, date_to=now), None),
        "tags": (rt.RetrieveFilters(call_tags=["a"]), None),
    }[label]


def test_choose_dense_mode_truth_table(monkeypatch):
    for row in GOLD["choose_dense_mode"]:
        monkeypatch.setattr(settings, "embeddings_exact_scan_threshold", row["threshold"])
        filters, call_ids = _scope(row["scope"])
        assert rt._choose_dense_mode(row["rows"], filters, call_ids) == row["mode"], row
        assert rt._dense_has_scoping(filters, call_ids) == row["has_scoping"], row


def test_choose_dense_mode_reference_unit_cases(monkeypatch):
    # /root/reference/tests/unit/test_retrieve_planner.py, same four cases
    monkeypatch.setattr(settings, "embeddings_exact_scan_threshold", 2000)
    f = rt.RetrieveFilters(call_ids=[uuid4()])
    assert rt._choose_dense_mode(estimated_rows=200, filters=f, call_ids=f.call_ids) == "exact"
    f = rt.RetrieveFilters(date_from=datetime.now(timezone.utc), date_to=datetime.now(timezone.utc))
    assert rt._choose_dense_mode(estimated_rows=5000, filters=f, call_ids=None) == "ann"
    monkeypatch.setattr(settings, "embeddings_exact_scan_threshold", 5000)
    assert rt._choose_dense_mode(estimated_rows=100, filters=None, call_ids=None) == "ann"
    monkeypatch.setattr(settings, "embeddings_exact_scan_threshold", 10)
    assert rt._choose_dense_mode(estimated_rows=0, filters=None, call_ids=None) == "exact"


# ---- S8 RRF ---------------------------------------------------------------------------------------
def test_rrf_merge_matches_reference():
    for case in GOLD["rrf_merge"]:
        lanes = {lane: [{"chunk_id": i, "tag": f"r{i}"} for i in ids] for lane, ids in case["lanes"]}
        merged = rt._rrf_merge(lanes, "chunk_id")
        assert [m[0]["chunk_id"] for m in merged] == case["order"]
        assert [sorted(m[1]) for m in merged] == case["hits"]
        assert [m[2] for m in merged] == case["scores"]  # same float operations, bit-identical


# ---- B1/B2 adaptive batching ------------------------------------------------------------------------
def test_infer_batch_size_limit_matches_reference():
    for case in GOLD["infer_batch_size_limit"]:
        assert embedding_pipeline.infer_batch_size_limit(case["message"]) == case["limit"], case


def test_embed_texts_adaptive_call_traces(monkeypatch):
    for case in GOLD["embed_texts_adaptive"]:
        calls = []

        def fake(texts, _c=case, _calls=calls):
            _calls.append(len(texts))
            if len(texts) > _c["limit"]:
                raise EmbeddingClientError(
                    f"inference request batch-size must be <= {_c['limit']}" if _c["hint"] else "backend busy")
            return EmbeddingResult(vectors=[[0.0] * 4 for _ in texts], model="m")

        monkeypatch.setattr(embedding_pipeline, "embed_texts", fake)
        res = embedding_pipeline._embed_texts_adaptive([f"t{i}" for i in range(case["n_texts"])],
                                                       batch_size=case["batch_size"])
        assert calls == case["calls"], case
        assert len(res.vectors) == case["n_vectors"] and res.model == case["model"]


def test_embed_texts_adaptive_single_row_failure_reraises(monkeypatch):
    def boom(texts):
        raise EmbeddingClientError("upstream unavailable")

    monkeypatch.setattr(embedding_pipeline, "embed_texts", boom)
    with pytest.raises(EmbeddingClientError):
        embedding_pipeline._embed_texts_adaptive(["only-one"], batch_size=4)


# ---- C1..C6 client ----------------------------------------------------------------------------------
class _FakeEncoder:
    def __init__(self, dim, model="fake-native"):
        self.dim, self.model, self.calls = dim, model, []

    def encode(self, texts):
        self.calls.append(list(texts))
        return [[float(len(t))] + [0.0] * (self.dim - 1) for t in texts], self.model


def test_client_error_strings_match_reference(monkeypatch):
    want = GOLD["client_errors"]
    monkeypatch.setattr(settings, "embeddings_base_url", "")
    with pytest.raises(EmbeddingClientError) as e:
        embeddings.embed_texts(["hello"])
    assert str(e.value) == want["not_configured"]["message"]
    with pytest.raises(EmbeddingClientError) as e:
        embeddings._validate_texts(["", "  ", None])
    assert str(e.value) == want["no_texts"]["message"]
    monkeypatch.setattr(settings, "embeddings_dim", 4)
    with pytest.raises(EmbeddingClientError) as e:
        embeddings._validate_vectors([[0.1, 0.2, 0.3, 0.4], [0.1, 0.2]])
    assert str(e.value) == want["bad_dim"]["message"]
    with pytest.raises(EmbeddingClientError) as e:
        embeddings.embed_texts_batched(["a"], batch_size=-1)
    assert str(e.value) == want["batch_zero"]["message"]
    assert embeddings._validate_texts(["  a ", "", "b", 3, None, " c"]) == want["validate_texts_keeps"]


class _Resp:
    def __init__(self, status, body):
        self.status_code, self._body, self.text = status, body, str(body)

    def json(self):
        return self._body


class _Client:
    def __init__(self, resp, log):
        self._resp, self._log = resp, log

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return None

    def post(self, url, json):
        self._log.append({"url": url, "payload": json})
        return self._resp(json) if callable(self._resp) else self._resp


def test_http_gateway_path_speaks_the_reference_protocol(monkeypatch):
    import httpx
    want = GOLD["client_errors"]
    log = []
    monkeypatch.setattr(settings, "embeddings_base_url", "http://embed.local/")
    monkeypatch.setattr(settings, "embeddings_dim", 2)
    for name, resp in (("http_500", _Resp(500, {"detail": "x" * 500})),
                       ("missing_list", _Resp(200, {"model": "m"})),
                       ("count_mismatch", _Resp(200, {"embeddings": [[1.0, 2.0]], "model": "m"}))):
        monkeypatch.setattr(httpx, "Client", lambda *a, _r=resp, **k: _Client(_r, log))
        with pytest.raises(EmbeddingClientError) as e:
            embeddings.embed_texts(["a", "b"])
        assert str(e.value) == want[name]["message"], name
    assert log[0]["url"] == "http://embed.local/embed"
    assert log[0]["payload"] == {"texts": ["a", "b"], "model": settings.embeddings_model_id}


def test_embed_texts_batched_splits_requests(monkeypatch):
    # /root/reference/tests/unit/test_embeddings_client.py::test_embed_texts_batched_splits_requests
    enc = _FakeEncoder(3)
    monkeypatch.setattr(settings, "embeddings_base_url", "native")
    monkeypatch.setattr(settings, "embeddings_dim", 3)
    embeddings.set_encoder(enc)
    try:
        res = embeddings.embed_texts_batched(["a", "b", "c", "d", "e"], batch_size=2)
    finally:
        embeddings.set_encoder(None)
    assert [len(c) for c in enc.calls] == [2, 2, 1]
    assert len(res.vectors) == 5 and res.model == "fake-native"


def test_native_backend_validates_dim_and_wraps_errors(monkeypatch):
    monkeypatch.setattr(settings, "embeddings_base_url", "native://qwen3")
    monkeypatch.setattr(settings, "embeddings_dim", 4)
    with pytest.raises(EmbeddingClientError, match="not loaded"):
        embeddings.embed_texts(["x"])
    embeddings.set_encoder(_FakeEncoder(3))
    try:
        with pytest.raises(EmbeddingClientError, match="embedding 0 has dim 3; expected 4"):
            embeddings.embed_texts(["x"])

        class Boom:
            def encode(self, texts):
                raise MemoryError("HIP out of memory")

        embeddings.set_encoder(Boom())
        with pytest.raises(EmbeddingClientError, match="native embedding encoder failed"):
            embeddings.embed_texts(["x"])
    finally:
        embeddings.set_encoder(None)


# ---- B6/B7 backfill ---------------------------------------------------------------------------------
def test_backfill_guards_match_reference(monkeypatch):
    want = GOLD["backfill_guards"]
    monkeypatch.setattr(embedding_pipeline, "embeddings_enabled", lambda: False)
    with pytest.raises(RuntimeError) as e:
        embedding_pipeline.run_embedding_backfill(batch_size=8)
    assert str(e.value) == want["disabled"]
    monkeypatch.setattr(embedding_pipeline, "embeddings_enabled", lambda: True)
    with pytest.raises(RuntimeError) as e:
        embedding_pipeline.run_embedding_backfill(batch_size=0)
    assert str(e.value) == want["batch_zero"]
    monkeypatch.setattr(settings, "embeddings_dim", 0)
    with pytest.raises(RuntimeError) as e:
        embedding_pipeline.run_embedding_backfill(batch_size=8)
    assert str(e.value) == want["dim_zero"]
    assert [vars(s) for s in embedding_pipeline.TABLE_SPECS] == GOLD["table_specs"]


def _tables():
    c1, c2 = UUID(int=1), UUID(int=2)
    return c1, c2, {
        "chunks": {
            3: {"call_id": c1, "text": "third", "embedding": None},
            1: {"call_id": c1, "text": "first", "embedding": None},
            2: {"call_id": c2, "text": "   ", "embedding": None},      # blank: never fetched
            4: {"call_id": c2, "text": "done", "embedding": [1.0, 0.0]},  # already embedded
            5: {"call_id": c2, "text": "fifth", "embedding": None},
        },
        "artifact_chunks": {7: {"call_id": c2, "text": "artifact", "embedding": None}},
    }


def test_run_embedding_backfill_end_to_end(monkeypatch, capsys):
    c1, c2, tables = _tables()
    store = embedding_pipeline.InMemoryStore(tables)
    enc = _FakeEncoder(2)
    monkeypatch.setattr(settings, "embeddings_base_url", "native")
    monkeypatch.setattr(settings, "embeddings_dim", 2)
    monkeypatch.setattr(settings, "embeddings_batch_size", 2)
    embeddings.set_encoder(enc)
    embedding_pipeline.set_store(store)
    try:
        summary = embedding_pipeline.run_embedding_backfill(batch_size=2, source="embed_backfill")
        assert summary.rows_updated == 4 and summary.per_table == {"chunks": 3, "artifact_chunks": 1}
        assert summary.calls_touched == 2 and summary.ingestion_runs_inserted == 2
        assert summary.model_used == "fake-native"
        assert enc.calls == [["first", "third"], ["fifth"], ["artifact"]]  # ORDER BY id, LIMIT batch
        assert tables["chunks"][2]["embedding"] is None and tables["chunks"][4]["embedding"] == [1.0, 0.0]
        cfg = json.loads(store.runs[0]["embedding_config"])
        assert cfg["enabled"] is True and cfg["dim"] == 2 and cfg["source"] == "embed_backfill"
        assert [r["call_id"] for r in store.runs] == sorted([c1, c2], key=str)
        # resumable: nothing left, second run is a no-op apart from the (empty) audit
        again = embedding_pipeline.run_embedding_backfill(batch_size=2)
        assert again.rows_updated == 0 and again.ingestion_runs_inserted == 0
        # per-call scope (ingest auto-embed path, ingest_fs.py:816)
        tables["chunks"][9] = {"call_id": c1, "text": "late", "embedding": None}
        tables["chunks"][10] = {"call_id": c2, "text": "other call", "embedding": None}
        scoped = embedding_pipeline.run_embedding_backfill(batch_size=8, call_id=c1, source="ingest_auto_embed")
        assert scoped.rows_updated == 1 and tables["chunks"][10]["embedding"] is None

        from cadence_rag_amd.scripts import embed_backfill
        embed_backfill.main()
        out = capsys.readouterr().out.strip().splitlines()
        assert out[0] == "[embed_backfill] finished table=chunks updated=1"
        assert out[1] == "[embed_backfill] finished table=artifact_chunks updated=0"
        # model = the LAST table's model; an idle last table reports the configured id (the
        # reference's loop does the same: embedding_pipeline.py:259-268)
        assert out[2] == ("[embed_backfill] complete rows_updated=1 calls_touched=1 "
                          f"ingestion_runs_inserted=1 model={settings.embeddings_model_id}")
    finally:
        embeddings.set_encoder(None)
        embedding_pipeline.set_store(None)


def test_update_embeddings_rejects_length_mismatch():
    embedding_pipeline.set_store(embedding_pipeline.InMemoryStore({"chunks": {}}))
    try:
        with pytest.raises(RuntimeError, match="row/vector mismatch for chunks: 1 rows vs 0 vectors"):
            embedding_pipeline._update_embeddings(
                embedding_pipeline.TABLE_SPECS[0],
                [embedding_pipeline.PendingRow(1, UUID(int=1), "x")], [])
    finally:
        embedding_pipeline.set_store(None)


# ---- B8 auto-embed hook (reference tests/unit/test_ingest_fs.py:130-195, same five cases) ------------
def test_auto_embed_hook_statuses(monkeypatch):
    from types import SimpleNamespace
    from cadence_rag_amd import ingest_hook
    monkeypatch.setattr(ingest_hook.settings, "ingest_auto_embed_on_success", False)
    assert ingest_hook._auto_embed_call_if_configured(uuid4()) == {"status": "skipped", "reason": "disabled"}
    monkeypatch.setattr(ingest_hook.settings, "ingest_auto_embed_on_success", True)
    monkeypatch.setattr(ingest_hook, "embeddings_enabled", lambda: False)
    assert ingest_hook._auto_embed_call_if_configured(uuid4()) == {"status": "skipped",
                                                                     "reason": "embeddings_not_configured"}
    monkeypatch.setattr(ingest_hook, "embeddings_enabled", lambda: True)
    monkeypatch.setattr(ingest_hook.settings, "ingest_auto_embed_fail_on_error", False)
    summary = SimpleNamespace(rows_updated=7, calls_touched=1, ingestion_runs_inserted=1,
                              model_used="Qwen/Qwen3-Embedding-4B", per_table={"chunks": 4, "artifact_chunks": 3})
    seen = {}
    monkeypatch.setattr(ingest_hook, "run_embedding_backfill", lambda **kw: seen.update(kw) or summary)
    cid = uuid4()
    res = ingest_hook._auto_embed_call_if_configured(cid)
    assert res == {"status": "ok", "rows_updated": 7, "calls_touched": 1, "model_used": "Qwen/Qwen3-Embedding-4B",
                   "ingestion_runs_inserted": 1}
    assert seen["call_id"] == cid and seen["source"] == "ingest_auto_embed" and seen["batch_size"] >= 1

    def boom(**kw):
        raise EmbeddingClientError("service timeout")

    monkeypatch.setattr(ingest_hook, "run_embedding_backfill", boom)
    res = ingest_hook._auto_embed_call_if_configured(uuid4())
    assert res["status"] == "error" and "service timeout" in res["error"]
    monkeypatch.setattr(ingest_hook.settings, "ingest_auto_embed_fail_on_error", True)
    with pytest.raises(EmbeddingClientError):
        ingest_hook._auto_embed_call_if_configured(uuid4())


# ---- S7 _resolve_call_ids ---------------------------------------------------------------------------
def test_resolve_call_ids_semantics():
    a, b, c = UUID(int=1), UUID(int=2), UUID(int=3)
    calls = [{"call_id": a, "external_id": "X", "external_source": "crm"},
             {"call_id": b, "external_id": "X", "external_source": None},
             {"call_id": c, "external_id": "Y", "external_source": "crm"}]
    assert rt._resolve_call_ids(calls, None) is None
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters()) is None
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters(call_ids=[c, a])) == sorted([a, c], key=str)
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters(external_id="X")) == sorted([a, b], key=str)
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters(external_id="X", external_source="crm")) == [a]
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters(external_id="X", call_ids=[b, c])) == [b]
    assert rt._resolve_call_ids(calls, rt.RetrieveFilters(external_id="nope")) == []


# ---- S9: retrieve_evidence orchestration vs the reference's own responses ---------------------
def _load_retrieve_goldens():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).parent / "golden" / "reference_retrieve_evidence.json").read_text())


class _ReplayBackend:
    """Serves the canned lane rows the goldens were captured with (the reference's SQL helpers were
    replaced by the same rows in tests/golden/make_goldens.py)."""

    def __init__(self, lanes, scenario):
        from uuid import UUID
        self.lanes, self.sc = lanes, scenario
        self.call_ids = [UUID(c) for c in scenario["call_ids"]] if scenario.get("call_ids") else None
        self.seen = []

    def resolve_call_ids(self, filters): return self.call_ids
    def fetch_chunks_bm25(self, q, f, c, k): return [dict(r) for r in self.lanes["bm25_chunks"]][:k]
    def fetch_artifacts_bm25(self, q, f, c, k): return [dict(r) for r in self.lanes["bm25_artifacts"]][:k]
    def fetch_chunks_tech(self, t, f, c, k): return [dict(r) for r in self.lanes["tech_chunks"]][:k] if t else []
    def fetch_artifacts_tech(self, t, f, c, k): return [dict(r) for r in self.lanes["tech_artifacts"]][:k] if t else []
    def estimate_dense_candidates(self, table, f, c): return self.sc["candidates"][table]

    def fetch_chunks_dense(self, e, f, c, mode, k):
        self.seen.append(("chunks", mode, k, e[:12]))
        return [dict(r) for r in self.lanes["dense_chunks"]][:k]

    def fetch_artifacts_dense(self, e, f, c, mode, k):
        self.seen.append(("artifact_chunks", mode, k, e[:12]))
        return [dict(r) for r in self.lanes["dense_artifacts"]][:k]


@pytest.mark.parametrize("idx", range(10))
def test_retrieve_evidence_matches_reference_responses(monkeypatch, idx):
    from uuid import UUID

    from cadence_rag_amd import embeddings, retrieve
    gold = _load_retrieve_goldens()
    sc = gold["scenarios"][idx]
    monkeypatch.setattr(embeddings, "embeddings_enabled", lambda: sc["dense"] != "off")

    def fake_embed(texts):
        if sc["dense"] == "error":
            raise embeddings.EmbeddingClientError("embedding request failed: connection refused")
        return embeddings.EmbeddingResult(vectors=[[0.25] * 1024 for _ in texts], model="Qwen/Qwen3-Embedding-4B")

    monkeypatch.setattr(embeddings, "embed_texts", fake_embed)
    pl = dict(sc["payload"])
    if "filters" in pl:
        f = dict(pl["filters"])
        if f.get("call_ids"):
            f["call_ids"] = [UUID(c) for c in f["call_ids"]]
        pl["filters"] = retrieve.RetrieveFilters(**f)
    if "budget" in pl:
        pl["budget"] = retrieve.Budget(**pl["budget"])
    be = _ReplayBackend(gold["lanes"], sc)
    resp = retrieve.retrieve_evidence(retrieve.RetrieveRequest(**pl), be)
    assert UUID(resp.pop("query_id"))
    assert resp == sc["response"], sc["name"]
    if sc["dense"] == "ok" and sc["payload"]["query"].strip():
        # the dense helpers got the .10g vector literal and the planner's mode, k = 50 / 10
        assert [s[0] for s in be.seen] == ["chunks", "artifact_chunks"]
        assert be.seen[0][2] == 50 and be.seen[1][2] == 10 and be.seen[0][3].startswith("[0.25,0.25")


def test_retrieve_evidence_needs_a_backend():
    from cadence_rag_amd import retrieve
    retrieve.set_backend(None)
    with pytest.raises(RuntimeError, match="no backend"):
        retrieve.retrieve_evidence(retrieve.RetrieveRequest(query="x"))


# ---- device-resident backfill path (embed_texts_device / DeviceSinkStore): same guards, no float lists --------
class _FakeDeviceEncoder:
    """encode_device protocol with CPU tensors (the contract is 'a float32 [n, dim] tensor', wherever it lives)."""

    def __init__(self, dim=1024, limit=None):
        self.dim, self.limit, self.calls = dim, limit, []

    def encode_device(self, texts):
        import torch
        self.calls.append(len(texts))
        if self.limit is not None and len(texts) > self.limit:
            raise RuntimeError(f"batch-size must be <= {self.limit}")
        v = torch.arange(len(texts) * self.dim, dtype=torch.float32).reshape(len(texts), self.dim)
        return v, "fake-model"

    def encode(self, texts):
        raise AssertionError("the list-returning form must not be used on the device path")


def test_embed_texts_device_guards_and_shapes(monkeypatch):
    from cadence_rag_amd import embeddings
    monkeypatch.setattr(embeddings.settings, "embeddings_base_url", "")
    with pytest.raises(EmbeddingClientError, match="EMBEDDINGS_BASE_URL is not configured"):
        embeddings.embed_texts_device(["x"])
    monkeypatch.setattr(embeddings.settings, "embeddings_base_url", "http://gateway:8000")
    with pytest.raises(EmbeddingClientError, match="need the native encoder"):
        embeddings.embed_texts_device(["x"])
    monkeypatch.setattr(embeddings.settings, "embeddings_base_url", "native")
    monkeypatch.setattr(embeddings.settings, "embeddings_dim", 1024)
    embeddings.set_encoder(_FakeDeviceEncoder())
    try:
        with pytest.raises(EmbeddingClientError, match="at least one non-empty text"):
            embeddings.embed_texts_device(["", "  "])
        res = embeddings.embed_texts_device([" a ", "", "b"])
        assert tuple(res.vectors.shape) == (2, 1024) and res.model == "fake-model"
        embeddings.set_encoder(_FakeDeviceEncoder(dim=512))
        with pytest.raises(EmbeddingClientError, match="embedding 0 has dim 512; expected 1024"):
            embeddings.embed_texts_device(["a"])
    finally:
        embeddings.set_encoder(None)


def test_device_sink_store_backfill_and_adaptive_downshift(monkeypatch):
    """run_embedding_backfill over a DeviceSinkStore: tensors go to the sink, rows are marked, and the batch
    downshift of _embed_texts_adaptive (embedding_pipeline.py:88-118) works on the device path too."""
    from cadence_rag_amd import embedding_pipeline as ep, embeddings
    monkeypatch.setattr(embeddings.settings, "embeddings_base_url", "native")
    monkeypatch.setattr(embeddings.settings, "embeddings_dim", 1024)
    enc = _FakeDeviceEncoder(limit=2)
    embeddings.set_encoder(enc)

    class Sink:
        def __init__(self):
            self.batches = []

        def add(self, vectors, ids):
            self.batches.append((tuple(vectors.shape), list(ids)))

    sink = Sink()
    tables = {"chunks": {i: {"call_id": UUID(int=1), "text": f"t{i}", "embedding": None} for i in range(5)},
              "artifact_chunks": {}}
    ep.set_store(ep.DeviceSinkStore(tables, sinks={"chunks": sink}))
    try:
        summary = ep.run_embedding_backfill(batch_size=5)
        assert summary.rows_updated == 5 and summary.per_table == {"chunks": 5, "artifact_chunks": 0}
        assert enc.calls == [5, 2, 2, 1]                      # the reference's downshift trace
        assert sink.batches == [((5, 1024), [0, 1, 2, 3, 4])]  # one tensor per fetched batch
        assert all(r["embedding"] == "hbm" for r in tables["chunks"].values())
        ep.set_store(ep.DeviceSinkStore({"chunks": {1: {"call_id": UUID(int=1), "text": "x", "embedding": None}},
                                         "artifact_chunks": {}}, sinks={}))
        with pytest.raises(RuntimeError, match="no sink"):
            ep.run_embedding_backfill(batch_size=5)
    finally:
        ep.set_store(None)
        embeddings.set_encoder(None)


def test_bench_synthetic_rows_do_not_depend_on_the_shard(monkeypatch):
    """bench.py defines the synthetic corpus chunk-wise from the seed: a rank that generates rows [lo, hi) gets
    the rows a single-GPU run sees there (N > 1 shards ONE corpus)."""
    import importlib.util
    import sys as _sys
    from pathlib import Path
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "SYNTH_CHUNK", 1000)
    cpu = torch.device("cpu")
    whole = bench.synth_rows(0, 3500, 7, cpu)
    assert torch.equal(bench.synth_rows(900, 2100, 7, cpu), whole[900:2100])
    assert torch.equal(bench.synth_rows(3000, 3500, 7, cpu), whole[3000:3500])
    assert torch.allclose(whole.norm(dim=1), torch.ones(3500), atol=1e-5)
    assert not torch.equal(bench.synth_rows(0, 10, 8, cpu), whole[:10])
