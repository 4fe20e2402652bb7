#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference itself (run in the build container only).

Imports the reference's pure-Python host logic from /root/reference (read-only) and records
input/output VALUES for the functions on the dense path's boundary.  The
reference needs `pydantic_settings`, which this image lacks: as SURVEY.md 8c prescribes, a
small in-memory stand-in for BaseSettings/SettingsConfigDict is installed in sys.modules for
the duration of this script, and DATABASE_URL=sqlite:// keeps app.db's create_engine from
needing psycopg.  Only data is written; no reference source text is copied.

The dense ARITHMETIC (pgvector, the embedding gateway) is not in /root/reference and cannot be
run: dense_scan_*.npz are generated from oracle/exact_scan.c (fp64 mode) and are the build's
own pin ("parity unpinned" by the reference — see DESIGN.md).
"""
from __future__ import annotations

import json
import os
import sys
import types
from datetime import datetime, timezone
from pathlib import Path
from uuid import UUID

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")


def _install_settings_stub() -> None:
    mod = types.ModuleType("pydantic_settings")

    class BaseSettings:  # reads class-level defaults; env overrides by upper-cased field name
        def __init__(self, **kw):
            for name, default in vars(type(self)).items():
                if name.startswith("_") or callable(default) or name == "model_config":
                    continue
                raw = os.environ.get(name.upper())
                value = default
                if raw is not None:
                    value = type(default)(raw) if not isinstance(default, bool) else raw.lower() in ("1", "true", "yes")
                setattr(self, name, kw.get(name, value))

    def SettingsConfigDict(**kw):
        return dict(kw)

    mod.BaseSettings = BaseSettings
    mod.SettingsConfigDict = SettingsConfigDict
    sys.modules["pydantic_settings"] = mod


def main() -> None:
    os.environ["DATABASE_URL"] = "sqlite://"
    os.environ["EMBEDDINGS_BASE_URL"] = ""
    _install_settings_stub()
    sys.path.insert(0, str(REF))
    import app.embedding_pipeline as pipeline  # noqa: E402
    import app.embeddings as embeddings  # noqa: E402
    import app.retrieve as retrieve  # noqa: E402
    from app.config import settings  # noqa: E402
    from app.ingest import extract_tech_tokens  # noqa: E402
    from app.schemas import RetrieveFilters  # noqa: E402

    sys.path.insert(0, str(REF / "eval"))
    from run_eval import compute_metrics  # noqa: E402

    out = {}

    # --- S1 _vector_literal (retrieve.py:263-264) ------------------------------------------
    rng = np.random.default_rng(20240501)
    vecs = [
        rng.standard_normal(16).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 1e-45, -1e-45, 3.4028235e38, 1.17549435e-38, 0.1, 1 / 3],
                 dtype=np.float32),
        (rng.standard_normal(8) * 1e-20).astype(np.float32),
    ]
    out["vector_literal"] = [
        {"values_f32_hex": [np.float32(v).tobytes().hex() for v in vec],
         "literal": retrieve._vector_literal([float(v) for v in vec])}
        for vec in vecs
    ]

    # --- S2 planner (retrieve.py:267-287) --------------------------------------------------
    some_call = UUID("11111111-2222-3333-4444-555555555555")
    now = datetime(2026, 1, 2, 3, 4, 5, tzinfo=timezone.utc)
    planner = []
    for threshold in (2000, 10, 0, -5):
        settings.embeddings_exact_scan_threshold = threshold
        for rows in (-1, 0, 1, 10, 11, 200, 2000, 2001, 5000):
            for label, filters, call_ids in (
                ("none", None, None),
                ("empty_filters", RetrieveFilters(), None),
                ("call_ids", RetrieveFilters(call_ids=[some_call]), [some_call]),
                ("resolved_empty", RetrieveFilters(external_id="x"), []),
                ("dates", RetrieveFilters(date_from=now, date_to=now), None),
                ("tags", RetrieveFilters(call_tags=["a"]), None),
            ):
                planner.append({"threshold": threshold, "rows": rows, "scope": label,
                                "mode": retrieve._choose_dense_mode(rows, filters, call_ids),
                                "has_scoping": retrieve._dense_has_scoping(filters, call_ids)})
    settings.embeddings_exact_scan_threshold = 2000
    out["choose_dense_mode"] = planner

    # --- S8 _rrf_merge (retrieve.py:245-260) -----------------------------------------------
    def rows(ids):
        return [{"chunk_id": i, "tag": f"r{i}"} for i in ids]

    rrf_cases = []
    for lanes in (
        {"bm25": rows([1, 2, 3]), "tech_tokens": rows([3, 4]), "dense": rows([2, 5, 1])},
        {"bm25": rows([7]), "tech_tokens": rows([8]), "dense": rows([9])},  # three-way tie
        {"bm25": [], "tech_tokens": [], "dense": rows([4, 3, 2, 1])},
        {"bm25": rows([1, 2]), "dense": rows([2, 1])},  # exact tie between two keys
    ):
        merged = retrieve._rrf_merge(lanes, "chunk_id")
        # lanes as an ordered list: insertion order decides ties and JSON objects get key-sorted
        rrf_cases.append({"lanes": [[k, [r["chunk_id"] for r in v]] for k, v in lanes.items()],
                          "order": [m[0]["chunk_id"] for m in merged],
                          "hits": [sorted(m[1]) for m in merged],
                          "scores": [m[2] for m in merged]})
    out["rrf_merge"] = rrf_cases

    # --- B1 infer_batch_size_limit (embedding_pipeline.py:71-85) ---------------------------
    msgs = [
        "Triton infer failed: [400] inference request batch-size must be <= 8 for 'qwen3_embed_4b_onnx'",
        "inference request batch-size must be <= 2",
        "maximum batch size is 16",
        "max batch-size: 4",
        "Max Batch Size exceeded; limit 0",
        "batch size must be <= 0",
        "upstream unavailable", "", "   ", "batch-size must be <= abc",
    ]
    out["infer_batch_size_limit"] = [{"message": m, "limit": pipeline.infer_batch_size_limit(m)} for m in msgs]

    # --- B2 _embed_texts_adaptive call-size traces (embedding_pipeline.py:88-118) -----------
    traces = []
    for n_texts, batch, limit, with_hint in ((5, 5, 2, True), (5, 5, 2, False), (9, 8, 3, True),
                                            (7, 32, 1, False), (4, 2, 8, True)):
        calls = []

        def fake(texts, _limit=limit, _hint=with_hint, _calls=calls):
            _calls.append(len(texts))
            if len(texts) > _limit:
                raise embeddings.EmbeddingClientError(
                    f"inference request batch-size must be <= {_limit}" if _hint else "backend busy")
            return embeddings.EmbeddingResult(vectors=[[0.0] * 4 for _ in texts], model="m")

        pipeline.embed_texts = fake
        res = pipeline._embed_texts_adaptive([f"t{i}" for i in range(n_texts)], batch_size=batch)
        traces.append({"n_texts": n_texts, "batch_size": batch, "limit": limit, "hint": with_hint,
                       "calls": calls, "n_vectors": len(res.vectors), "model": res.model})
    out["embed_texts_adaptive"] = traces

    # --- C2/C3/C4 error strings of the client (embeddings.py:29-100) ------------------------
    errors = {}

    def capture(name, fn):
        try:
            fn()
            errors[name] = None
        except Exception as exc:  # noqa: BLE001
            errors[name] = {"type": type(exc).__name__, "message": str(exc)}

    settings.embeddings_base_url = ""
    capture("not_configured", lambda: embeddings.embed_texts(["hello"]))
    capture("no_texts", lambda: embeddings._validate_texts(["", "  ", None]))  # type: ignore[list-item]
    settings.embeddings_dim = 4
    capture("bad_dim", lambda: embeddings._validate_vectors([[0.1, 0.2, 0.3, 0.4], [0.1, 0.2]]))
    capture("batch_zero", lambda: embeddings.embed_texts_batched(["a"], batch_size=-1))
    settings.embeddings_dim = 1024
    errors["validate_texts_keeps"] = embeddings._validate_texts(["  a ", "", "b", 3, None, " c"])  # type: ignore[list-item]

    class _Resp:
        def __init__(self, status, body):
            self.status_code, self._body, self.text = status, body, str(body)

        def json(self):
            return self._body

    class _Client:
        def __init__(self, resp):
            self._resp = resp

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return None

        def post(self, url, json):
            return self._resp

    settings.embeddings_base_url = "http://embed.local/"
    settings.embeddings_dim = 2
    for name, resp in (("http_500", _Resp(500, {"detail": "x" * 500})),
                       ("missing_list", _Resp(200, {"model": "m"})),
                       ("count_mismatch", _Resp(200, {"embeddings": [[1.0, 2.0]], "model": "m"}))):
        embeddings.httpx.Client = lambda *a, _r=resp, **k: _Client(_r)
        capture(name, lambda: embeddings.embed_texts(["a", "b"]))
    settings.embeddings_base_url = ""
    settings.embeddings_dim = 1024
    out["client_errors"] = errors

    # --- backfill guards (embedding_pipeline.py:247-252) ------------------------------------
    guards = {}
    pipeline.embeddings_enabled = lambda: False
    try:
        pipeline.run_embedding_backfill(batch_size=8)
    except RuntimeError as exc:
        guards["disabled"] = str(exc)
    pipeline.embeddings_enabled = lambda: True
    try:
        pipeline.run_embedding_backfill(batch_size=0)
    except RuntimeError as exc:
        guards["batch_zero"] = str(exc)
    settings.embeddings_dim = 0
    try:
        pipeline.run_embedding_backfill(batch_size=8)
    except RuntimeError as exc:
        guards["dim_zero"] = str(exc)
    settings.embeddings_dim = 1024
    out["backfill_guards"] = guards
    out["table_specs"] = [vars(s) for s in pipeline.TABLE_SPECS]

    # --- tech tokens (ingest.py:141-160) — "next" row, recorded for later ------------------
    sentences = [
        "Where did we discuss ECONNRESET in api-gateway? ticket ABC-123 v1.2.3",
        "See https://example.com/a/b?c=1 and 10.25.0.50, ORA-12345, HTTP 502, commit deadbeef1234",
        "Dell vs Lenovo bake off for the object storage BOM; /etc/hosts and build 2.0",
        "nothing technical here",
    ]
    out["extract_tech_tokens"] = [{"text": s, "tokens": extract_tech_tokens(s)} for s in sentences]

    # --- eval metrics (eval/run_eval.py:26-65) ---------------------------------------------
    gold = {"q1": ["chunk:1", "chunk:2"], "q2": ["chunk:9"], "q3": []}
    results = {"q1": ["chunk:2", "chunk:7", "chunk:1"], "q2": ["chunk:3", "chunk:4"]}
    out["compute_metrics"] = {"gold": gold, "results": results, "ks": [1, 2, 10],
                              "metrics": compute_metrics(gold, results, [1, 2, 10])}

    # --- S9 retrieve_evidence (retrieve.py:392-688): orchestration, RRF, evidence packing ------
    _retrieve_evidence_goldens(retrieve, embeddings)

    (HERE / "reference_host_logic.json").write_text(json.dumps(out, indent=1, sort_keys=True, default=str) + "\n")
    print("wrote", HERE / "reference_host_logic.json")

    # --- dense scan fixtures from the oracle (fp64 truth) ----------------------------------
    sys.path.insert(0, str(REPO))
    import oracle

    def unit(a):
        return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)

    rng = np.random.default_rng(1234)
    corpus = unit(rng.standard_normal((4096, 1024)))
    queries = rng.standard_normal((8, 1024)).astype(np.float32)
    ids, scores, counts = oracle.exact_topk(queries, corpus, 50, mode=oracle.F64)
    np.savez_compressed(HERE / "dense_scan_4096.npz", seed=1234, n=4096, k=50,
                        queries=queries, ids=ids.astype(np.int32), scores=scores)

    # small adversarial corpus: duplicates, zero rows, a NaN row, +-q rows, ragged tail (257 rows)
    rng = np.random.default_rng(4321)
    c2 = rng.standard_normal((257, 1024)).astype(np.float32)
    q2 = rng.standard_normal((3, 1024)).astype(np.float32)
    c2[5] = 0.0
    c2[17] = c2[3]
    c2[200] = c2[3]
    c2[64] = q2[0] * 3.0
    c2[65] = -q2[0]
    c2[100, 7] = np.nan
    c2[256] = q2[1]
    ids2, scores2, counts2 = oracle.exact_topk(q2, c2, 20, mode=oracle.F64)
    mask = rng.random((3, 257)) < 0.3
    ids3, scores3, counts3 = oracle.exact_topk(q2, c2, 20, mask=np.packbits(mask, axis=-1, bitorder="little"),
                                               mode=oracle.F64)
    np.savez_compressed(HERE / "dense_scan_257_edge.npz", corpus=c2, queries=q2, k=20,
                        ids=ids2.astype(np.int32), scores=scores2, counts=counts2,
                        mask=mask, masked_ids=ids3.astype(np.int32), masked_scores=scores3,
                        masked_counts=counts3)
    print("wrote dense fixtures")


def _retrieve_evidence_goldens(retrieve, embeddings) -> None:
    """Run the reference's retrieve_evidence with its SQL helpers replaced by canned lane rows and
    record (inputs, response) pairs.  Only values are stored."""
    from contextlib import contextmanager

    from app.schemas import Budget, RetrieveFilters, RetrieveRequest

    calls = ["7f0c1d3e-0000-4000-8000-00000000000%d" % i for i in range(5)]

    def chunk(cid, call, text, speaker="A", score=None):
        row = {"chunk_id": cid, "call_id": calls[call], "speaker": speaker, "start_ts_ms": cid * 1000,
               "end_ts_ms": cid * 1000 + 900, "text": text}
        if score is not None:
            row["score"] = score
        return row

    def art(aid, call, content, kind="summary", score=None):
        row = {"artifact_chunk_id": aid, "artifact_id": 100 + aid, "call_id": calls[call], "kind": kind,
               "content": content}
        if score is not None:
            row["score"] = score
        return row

    long_text = ("The rollout of build 2024.11.3 failed on node gpu-17 with ECC errors. " * 20).strip()
    lanes = {
        "bm25_chunks": [chunk(11, 0, long_text, score=9.5), chunk(12, 0, "second chunk of call zero", score=7.25),
                        chunk(13, 0, "third chunk of call zero", score=7.0), chunk(21, 1, "call one talks about ERR-4521", score=6.5),
                        chunk(31, 2, "short", score=1.0)],
        "bm25_artifacts": [art(5, 0, long_text, score=4.0), art(6, 1, "decision: roll back", kind="decisions", score=3.0),
                           art(7, 2, "action: file ticket OPS-99", kind="action_items", score=2.0)],
        "tech_chunks": [chunk(21, 1, "call one talks about ERR-4521"), chunk(41, 3, "ERR-4521 seen again")],
        "tech_artifacts": [art(7, 2, "action: file ticket OPS-99", kind="action_items")],
        "dense_chunks": [chunk(12, 0, "second chunk of call zero", score=0.91), chunk(41, 3, "ERR-4521 seen again", score=0.9),
                         chunk(51, 4, "unrelated but close", score=0.5), chunk(11, 0, long_text, score=0.4)],
        "dense_artifacts": [art(6, 1, "decision: roll back", kind="decisions", score=0.8), art(8, 3, "summary of call three", score=0.7)],
    }
    scenarios = [
        {"name": "evidence_pack_dense", "payload": {"query": "  why did ERR-4521 happen on gpu-17?  "},
         "dense": "ok", "candidates": {"chunks": 120000, "artifact_chunks": 900}},
        {"name": "ids_only_debug", "payload": {"query": "ERR-4521 rollback", "return_style": "ids_only", "debug": True},
         "dense": "ok", "candidates": {"chunks": 120000, "artifact_chunks": 900}},
        {"name": "evidence_pack_debug_scoped_exact", "payload": {"query": "ERR-4521 rollback", "debug": True,
                                                                  "filters": {"call_ids": [calls[0], calls[1]]}},
         "dense": "ok", "candidates": {"chunks": 400, "artifact_chunks": 0}, "call_ids": [calls[0], calls[1]]},
        {"name": "embedding_error_falls_back", "payload": {"query": "ERR-4521 rollback", "intent": "troubleshooting"},
         "dense": "error", "candidates": {"chunks": 5, "artifact_chunks": 5}},
        {"name": "dense_disabled", "payload": {"query": "ERR-4521 rollback"}, "dense": "off",
         "candidates": {"chunks": 5, "artifact_chunks": 5}},
        {"name": "small_budget", "payload": {"query": "ERR-4521 rollback", "budget": {"max_evidence_items": 3, "max_total_chars": 500}},
         "dense": "ok", "candidates": {"chunks": 10, "artifact_chunks": 10}},
        {"name": "one_item_budget", "payload": {"query": "ERR-4521 rollback", "budget": {"max_evidence_items": 1, "max_total_chars": 50}},
         "dense": "ok", "candidates": {"chunks": 10, "artifact_chunks": 10}},
        {"name": "empty_query_pack", "payload": {"query": "   "}, "dense": "ok", "candidates": {"chunks": 0, "artifact_chunks": 0}},
        {"name": "empty_query_ids", "payload": {"query": "", "return_style": "ids_only"}, "dense": "ok",
         "candidates": {"chunks": 0, "artifact_chunks": 0}},
        {"name": "no_tech_tokens_query", "payload": {"query": "what was decided about the budget", "return_style": "ids_only"},
         "dense": "ok", "candidates": {"chunks": 120000, "artifact_chunks": 120000}},
    ]

    class _Engine:
        @contextmanager
        def connect(self):
            yield object()

    saved = {name: getattr(retrieve, name) for name in (
        "engine", "_resolve_call_ids", "_fetch_chunks_bm25", "_fetch_artifacts_bm25", "_fetch_chunks_tech",
        "_fetch_artifacts_tech", "_estimate_dense_candidates", "_fetch_chunks_dense", "_fetch_artifacts_dense",
        "embeddings_enabled", "embed_texts")}
    results = []
    try:
        retrieve.engine = _Engine()
        for sc in scenarios:
            from uuid import UUID
            cids = [UUID(c) for c in sc["call_ids"]] if sc.get("call_ids") else None
            retrieve._resolve_call_ids = lambda conn, filters, _c=cids: _c
            retrieve._fetch_chunks_bm25 = lambda conn, q, f, c, k: [dict(r) for r in lanes["bm25_chunks"]][:k]
            retrieve._fetch_artifacts_bm25 = lambda conn, q, f, c, k: [dict(r) for r in lanes["bm25_artifacts"]][:k]
            retrieve._fetch_chunks_tech = lambda conn, t, f, c, k: ([dict(r) for r in lanes["tech_chunks"]][:k] if t else [])
            retrieve._fetch_artifacts_tech = lambda conn, t, f, c, k: ([dict(r) for r in lanes["tech_artifacts"]][:k] if t else [])
            retrieve._estimate_dense_candidates = lambda conn, table, f, c, _sc=sc: _sc["candidates"][table]
            retrieve._fetch_chunks_dense = lambda conn, e, f, c, m, k: [dict(r) for r in lanes["dense_chunks"]][:k]
            retrieve._fetch_artifacts_dense = lambda conn, e, f, c, m, k: [dict(r) for r in lanes["dense_artifacts"]][:k]
            retrieve.embeddings_enabled = lambda _sc=sc: _sc["dense"] != "off"

            def fake_embed(texts, _sc=sc):
                if _sc["dense"] == "error":
                    raise embeddings.EmbeddingClientError("embedding request failed: connection refused")
                return embeddings.EmbeddingResult(vectors=[[0.25] * 1024 for _ in texts], model="Qwen/Qwen3-Embedding-4B")

            retrieve.embed_texts = fake_embed
            pl = dict(sc["payload"])
            if "filters" in pl:
                pl["filters"] = RetrieveFilters(**pl["filters"])
            if "budget" in pl:
                pl["budget"] = Budget(**pl["budget"])
            resp = retrieve.retrieve_evidence(RetrieveRequest(**pl))
            resp.pop("query_id")
            results.append({"name": sc["name"], "payload": sc["payload"], "dense": sc["dense"],
                            "candidates": sc["candidates"], "call_ids": sc.get("call_ids"), "response": resp})
    finally:
        for name, val in saved.items():
            setattr(retrieve, name, val)
    (HERE / "reference_retrieve_evidence.json").write_text(
        json.dumps({"lanes": lanes, "scenarios": results}, indent=1, default=str) + "\n")
    print("wrote", HERE / "reference_retrieve_evidence.json")


if __name__ == "__main__":
    main()
