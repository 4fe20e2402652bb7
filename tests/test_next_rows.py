""""Next" rows of SURVEY.md 8(f): eval metrics/gate, pgvector interchange formats, tech tokens, the
HTTP /embed shim — host logic checked against values captured from the reference."""
import json
import struct
from pathlib import Path

import numpy as np
import pytest

from cadence_rag_amd import vector_io
from cadence_rag_amd.eval import compute_metrics, gate_failures
from cadence_rag_amd.eval import metrics as eval_cli
from cadence_rag_amd.tech_tokens import extract_tech_tokens

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "reference_host_logic.json").read_text())


def test_compute_metrics_matches_reference():
    g = GOLD["compute_metrics"]
    assert compute_metrics(g["gold"], g["results"], g["ks"]) == g["metrics"]
    assert compute_metrics({"q": []}, {}, [5]) == {"recall@5": 0.0, "mrr": 0.0, "ndcg@5": 0.0}


def test_regression_gate_cli(tmp_path, capsys):
    g = GOLD["compute_metrics"]
    (tmp_path / "gold.jsonl").write_text("\n".join(json.dumps({"query_id": q, "relevant_ids": r})
                                                   for q, r in g["gold"].items()) + "\n")
    (tmp_path / "res.jsonl").write_text("\n".join(json.dumps({"query_id": q, "retrieved_ids": r})
                                                  for q, r in g["results"].items()) + "\n\n")
    base = ["--gold", str(tmp_path / "gold.jsonl"), "--results", str(tmp_path / "res.jsonl")]
    eval_cli.main(base + ["--k", "1", "2", "10"])
    assert json.loads(capsys.readouterr().out) == g["metrics"]
    eval_cli.main(base + ["--gate", "--min-mrr", "0.5", "--min-recall-at", "10", "--min-recall", "0.5"])
    assert capsys.readouterr().out.rstrip().endswith("[regression_gate] PASS")
    with pytest.raises(SystemExit):
        eval_cli.main(base + ["--gate", "--min-mrr", "0.9"])
    assert "[regression_gate] FAIL" in capsys.readouterr().out
    assert gate_failures({"mrr": 0.4, "recall@20": 0.5, "ndcg@10": 0.1}, min_mrr=0.6, min_recall=0.8, min_ndcg=0.7) == [
        "mrr 0.4000 < 0.6000", "recall@20 0.5000 < 0.8000", "ndcg@10 0.1000 < 0.7000"]


def test_extract_tech_tokens_matches_reference():
    for case in GOLD["extract_tech_tokens"]:
        assert extract_tech_tokens(case["text"]) == case["tokens"], case["text"]


def test_pgvector_text_form_roundtrips_float32():
    for case in GOLD["vector_literal"]:
        vals = np.array([struct.unpack("<f", bytes.fromhex(h))[0] for h in case["values_f32_hex"]], dtype=np.float32)
        lit = vector_io.format_vector(vals)
        assert lit == case["literal"]
        assert vector_io.parse_vector(lit, len(vals)).tobytes() == vals.tobytes()  # bit-exact, -0.0 included
    rng = np.random.default_rng(0)
    m = rng.standard_normal((5, 1024)).astype(np.float32)
    back = vector_io.parse_vectors([vector_io.format_vector(r) for r in m], 1024)
    assert back.tobytes() == m.tobytes()
    with pytest.raises(ValueError):
        vector_io.parse_vector("[1,2,3]", 4)
    with pytest.raises(ValueError):
        vector_io.parse_vector("1,2,3")


def test_pgvector_binary_form():
    v = np.array([1.5, -2.25, 0.0, 3.4028235e38], dtype=np.float32)
    buf = vector_io.to_binary(v)
    assert buf[:4] == struct.pack(">hh", 4, 0) and len(buf) == 20
    assert buf[4:8] == struct.pack(">f", 1.5)
    assert np.array_equal(vector_io.from_binary(buf), v)
    with pytest.raises(ValueError):
        vector_io.from_binary(buf[:-1])
    assert vector_io.copy_rows_text([7], v[None])[0] == "7\t[1.5,-2.25,0,3.402823466e+38]"


def test_http_embed_shim_speaks_the_gateway_contract(monkeypatch):
    from fastapi.testclient import TestClient
    from cadence_rag_amd import embeddings, gateway
    from cadence_rag_amd.config import settings

    class Enc:
        def encode(self, texts):
            return [[float(len(t)), 0.0, 1.0] for t in texts], "Qwen/Qwen3-Embedding-4B"

    monkeypatch.setattr(settings, "embeddings_dim", 3)
    client = TestClient(gateway.app)
    assert client.get("/health").json()["status"] == "degraded"
    assert client.post("/embed", json={"texts": ["x"]}).status_code == 502
    embeddings.set_encoder(Enc())
    try:
        assert client.get("/health").json()["status"] == "ok"
        r = client.post("/embed", json={"texts": ["hello", "  ", "hi"], "model": "m"})
        assert r.status_code == 200
        assert r.json() == {"embeddings": [[5.0, 0.0, 1.0], [2.0, 0.0, 1.0]], "model": "m"}
        assert client.post("/embed", json={"texts": ["", " "]}).status_code == 400
        # and the reference-protocol client of this package can consume it (same JSON keys)
        body = client.post("/embed", json={"texts": ["abc"]}).json()
        assert set(body) == {"embeddings", "model"} and body["model"] == "Qwen/Qwen3-Embedding-4B"
    finally:
        embeddings.set_encoder(None)


def test_http_retrieve_route_serves_reference_responses(monkeypatch):
    """POST /retrieve (app/main.py:184-186) over retrieve_evidence: the JSON bodies equal the responses the
    reference produced for the same lane rows (tests/golden/reference_retrieve_evidence.json)."""
    import json
    from pathlib import Path

    from fastapi.testclient import TestClient
    from cadence_rag_amd import embeddings, gateway, retrieve
    from test_host_logic import _ReplayBackend

    gold = json.loads((Path(__file__).parent / "golden" / "reference_retrieve_evidence.json").read_text())
    client = TestClient(gateway.app)
    retrieve.set_backend(None)
    assert client.post("/retrieve", json={"query": "x"}).status_code == 503
    assert client.post("/retrieve", json={"query": "x", "intent": "nope"}).status_code == 422
    try:
        for sc in gold["scenarios"]:
            monkeypatch.setattr(embeddings, "embeddings_enabled", lambda sc=sc: sc["dense"] != "off")

            def fake_embed(texts, sc=sc):
                if sc["dense"] == "error":
                    raise embeddings.EmbeddingClientError("embedding request failed: connection refused")
                return embeddings.EmbeddingResult(vectors=[[0.25] * 1024 for _ in texts], model="Qwen/Qwen3-Embedding-4B")

            monkeypatch.setattr(embeddings, "embed_texts", fake_embed)
            retrieve.set_backend(_ReplayBackend(gold["lanes"], sc))
            r = client.post("/retrieve", json=sc["payload"])
            assert r.status_code == 200, sc["name"]
            body = r.json()
            body.pop("query_id")
            assert body == sc["response"], sc["name"]
    finally:
        retrieve.set_backend(None)
