"""Embedding backfill — same entry point and semantics as the reference's
app/embedding_pipeline.py (/root/reference/app/embedding_pipeline.py:25-282):
run_embedding_backfill(*, batch_size, call_id=None, source="embed_backfill") -> BackfillSummary,
the adaptive batch downshift of _embed_texts_adaptive, infer_batch_size_limit and the guard
messages (all pinned by tests/golden/reference_host_logic.json).

The reference hard-wires Postgres (SELECT ... WHERE embedding IS NULL / per-row UPDATE / INSERT
INTO ingestion_runs).  Here the three SQL steps are a BackfillStore interface, so the same loop
drives an in-memory store (tests), a synthetic source (bench config 4) or a SQL store written by
the reference's maintainers, and every embedded batch can also be appended to
the GPU-resident DenseIndex (sink) instead of being serialised as 15 KB text literals per row.
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass
from datetime import datetime, timezone
from typing import Dict, Iterable, List, Optional, Protocol, Sequence, Set, Tuple
from uuid import UUID

from . import embeddings as _emb
from .config import settings
from .embeddings import EmbeddingClientError, EmbeddingResult, embed_texts, embeddings_enabled  # noqa: F401

PIPELINE_VERSION = "v2"                 # /root/reference/app/ingest.py:19
NER_CONFIG_DISABLED = {"enabled": False}  # /root/reference/app/ingest.py:21
EMBEDDING_MODE = "mi355x_native_v1"     # the reference records "http_backfill_v1"


@dataclass(frozen=True)
class PendingRow:
    row_id: int
    call_id: UUID
    content: str


@dataclass(frozen=True)
class TableSpec:
    table: str
    id_column: str
    text_column: str
    order_column: str


@dataclass(frozen=True)
class BackfillSummary:
    rows_updated: int
    calls_touched: int
    ingestion_runs_inserted: int
    model_used: str
    per_table: Dict[str, int]


TABLE_SPECS: Sequence[TableSpec] = (
    TableSpec(table="chunks", id_column="chunk_id", text_column="text", order_column="chunk_id"),
    TableSpec(table="artifact_chunks", id_column="artifact_chunk_id", text_column="content",
              order_column="artifact_chunk_id"),
)


class BackfillStore(Protocol):
    """The three storage steps of the reference loop (embedding_pipeline.py:121-214)."""

    def fetch_pending_rows(self, spec: TableSpec, limit: int, call_id: Optional[UUID]) -> List[PendingRow]:
        """Rows with no embedding and non-blank text, ascending id, at most `limit`."""

    def update_embeddings(self, spec: TableSpec, rows: Sequence[PendingRow],
                          vectors: Sequence[Sequence[float]]) -> None:
        """Persist one vector per row (all-or-nothing per batch)."""

    def record_embedding_runs(self, call_ids: Iterable[UUID], embedding_config: dict,
                              chunking_config: dict) -> int:
        """One audit row per distinct call; returns the number inserted."""


_store: Optional[BackfillStore] = None


def set_store(store: Optional[BackfillStore]) -> None:
    global _store
    _store = store


def _vector_literal(values: Sequence[float]) -> str:
    """pgvector text form, '.10g' per component: round-trips float32 exactly."""
    return "[" + ",".join(format(float(v), ".10g") for v in values) + "]"


def _now_utc_iso() -> str:
    return datetime.now(timezone.utc).isoformat()


_LIMIT_PATTERNS = (
    re.compile(r"batch[- ]size[^0-9]{0,40}<=\s*(\d+)", re.IGNORECASE),
    re.compile(r"max(?:imum)?\s+batch[- ]size[^0-9]{0,40}(\d+)", re.IGNORECASE),
)


def infer_batch_size_limit(error_message: str) -> Optional[int]:
    msg = (error_message or "").strip()
    if not msg:
        return None
    for pat in _LIMIT_PATTERNS:
        m = pat.search(msg)
        if m:
            n = int(m.group(1))
            if n > 0:
                return n
    return None


def _embed_texts_adaptive(texts: Sequence[str], batch_size: int, on_device: bool = False):
    """on_device: the vectors stay on the GPU (embeddings.embed_texts_device) and come back as ONE float32
    [n, dim] tensor in a DeviceEmbeddingResult; the downshift-and-retry rule is the same."""
    cleaned = [t.strip() for t in texts if isinstance(t, str) and t.strip()]
    if not cleaned:
        raise EmbeddingClientError("embedding request requires at least one non-empty text")
    step = max(1, int(batch_size))
    vectors: list = []
    model = settings.embeddings_model_id
    pos = 0
    while pos < len(cleaned):
        part = cleaned[pos:pos + step]
        try:
            res = _emb.embed_texts_device(part) if on_device else embed_texts(part)
        except EmbeddingClientError as exc:
            if len(part) <= 1:
                raise
            hinted = infer_batch_size_limit(str(exc))
            step = max(1, hinted) if (hinted is not None and hinted < len(part)) else max(1, len(part) // 2)
            continue  # retry the same position with the smaller batch
        if on_device:
            vectors.append(res.vectors)
        else:
            vectors.extend(res.vectors)
        model = res.model
        pos += len(part)
    if on_device:
        import torch
        return _emb.DeviceEmbeddingResult(vectors=vectors[0] if len(vectors) == 1 else torch.cat(vectors), model=model)
    return EmbeddingResult(vectors=vectors, model=model)


def _require_store() -> BackfillStore:
    if _store is None:
        raise RuntimeError("no BackfillStore configured (cadence_rag_amd.embedding_pipeline.set_store)")
    return _store


def _fetch_pending_rows(spec: TableSpec, limit: int, call_id: Optional[UUID]) -> List[PendingRow]:
    return list(_require_store().fetch_pending_rows(spec, limit, call_id))


def _update_embeddings(spec: TableSpec, rows: Sequence[PendingRow],
                       vectors: Sequence[Sequence[float]]) -> None:
    if len(rows) != len(vectors):
        raise RuntimeError(f"row/vector mismatch for {spec.table}: {len(rows)} rows vs {len(vectors)} vectors")
    _require_store().update_embeddings(spec, rows, vectors)


def _record_embedding_runs(call_ids: Iterable[UUID], model_id: str, dim: int, source: str) -> int:
    chunking_config = {"enabled": True, "mode": "existing_chunks", "source": source}
    embedding_config = {
        "enabled": True, "mode": EMBEDDING_MODE, "model_id": model_id, "dim": dim,
        "base_url": settings.embeddings_base_url, "timestamp": _now_utc_iso(), "source": source,
    }
    ordered = sorted(set(call_ids), key=str)
    return int(_require_store().record_embedding_runs(ordered, embedding_config, chunking_config))


def _backfill_table(spec: TableSpec, *, batch_size: int, call_id: Optional[UUID]) -> Tuple[int, Set[UUID], str]:
    updated = 0
    touched: Set[UUID] = set()
    model = settings.embeddings_model_id
    while True:
        batch = _fetch_pending_rows(spec, batch_size, call_id=call_id)
        if not batch:
            break
        # a store that keeps vectors in HBM takes them as a device tensor: no per-float Python objects at all
        on_device = bool(getattr(_require_store(), "device_vectors", False))
        res = _embed_texts_adaptive([r.content for r in batch], batch_size=batch_size, on_device=on_device)
        _update_embeddings(spec, batch, res.vectors)
        touched.update(r.call_id for r in batch)
        updated += len(batch)
        model = res.model
    return updated, touched, model


def run_embedding_backfill(*, batch_size: int, call_id: Optional[UUID] = None,
                           source: str = "embed_backfill") -> BackfillSummary:
    if not embeddings_enabled():
        raise RuntimeError("EMBEDDINGS_BASE_URL must be set to run embedding backfill")
    if settings.embeddings_dim <= 0:
        raise RuntimeError("EMBEDDINGS_DIM must be > 0")
    if batch_size <= 0:
        raise RuntimeError("EMBEDDINGS_BATCH_SIZE must be > 0")
    total = 0
    calls: Set[UUID] = set()
    model = settings.embeddings_model_id
    per_table: Dict[str, int] = {}
    for spec in TABLE_SPECS:
        n, touched, model = _backfill_table(spec, batch_size=batch_size, call_id=call_id)
        per_table[spec.table] = n
        total += n
        calls.update(touched)
    inserted = _record_embedding_runs(calls, model_id=model, dim=settings.embeddings_dim, source=source)
    return BackfillSummary(rows_updated=total, calls_touched=len(calls), ingestion_runs_inserted=inserted,
                           model_used=model, per_table=per_table)


# ------------------------------------------------------------------------------------------------
# stores
# ------------------------------------------------------------------------------------------------
class InMemoryStore:
    """Dict-backed BackfillStore: rows = {table: {row_id: {"call_id", "text", "embedding"}}}.
    Optionally mirrors every embedded batch into DenseIndex sinks {table: DenseIndex}, which is how
    the GPU lane is populated without the per-row text-literal round trip."""

    def __init__(self, tables: Dict[str, Dict[int, dict]], sinks: Optional[dict] = None) -> None:
        self.tables = tables
        self.sinks = sinks or {}
        self.runs: List[dict] = []

    def fetch_pending_rows(self, spec, limit, call_id):
        rows = []
        for rid in sorted(self.tables.get(spec.table, {})):
            r = self.tables[spec.table][rid]
            if r.get("embedding") is not None:
                continue
            text = r.get("text")
            if text is None or not str(text).strip():
                continue
            if call_id is not None and r["call_id"] != call_id:
                continue
            rows.append(PendingRow(row_id=rid, call_id=r["call_id"], content=text))
            if len(rows) >= limit:
                break
        return rows

    def update_embeddings(self, spec, rows, vectors):
        for row, vec in zip(rows, vectors):
            self.tables[spec.table][row.row_id]["embedding"] = list(vec)
        sink = self.sinks.get(spec.table)
        if sink is not None and rows:
            sink.add([list(v) for v in vectors], ids=[r.row_id for r in rows])

    def record_embedding_runs(self, call_ids, embedding_config, chunking_config):
        n = 0
        for cid in call_ids:
            self.runs.append({"call_id": cid, "pipeline_version": PIPELINE_VERSION,
                              "chunking_config": json.dumps(chunking_config),
                              "embedding_config": json.dumps(embedding_config),
                              "ner_config": json.dumps(NER_CONFIG_DISABLED)})
            n += 1
        return n


class DeviceSinkStore(InMemoryStore):
    """BackfillStore whose vectors never leave the GPU: run_embedding_backfill sees `device_vectors`, embeds with
    embeddings.embed_texts_device and hands update_embeddings ONE float32 [n, dim] CUDA tensor, which goes to
    the table's sink (a DenseIndex, or DenseTable.sink(...)) as a device pointer -> crag_index_add.  The row
    dictionaries only record that the row is embedded (and where).  This removes the reference's per-row
    15 KB text literal (embedding_pipeline.py:157-168) AND the host float lists from the write path."""

    device_vectors = True

    def update_embeddings(self, spec, rows, vectors):
        sink = self.sinks.get(spec.table)
        if sink is None:
            raise RuntimeError(f"DeviceSinkStore has no sink for table {spec.table!r}")
        if rows:
            sink.add(vectors, ids=[r.row_id for r in rows])
        for row in rows:
            self.tables[spec.table][row.row_id]["embedding"] = "hbm"
