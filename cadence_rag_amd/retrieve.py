"""Dense lane of /retrieve — the counterpart of the dense helpers in the reference's
app/retrieve.py (/root/reference/app/retrieve.py:245-389): _vector_literal, _dense_has_scoping,
_choose_dense_mode, _estimate_dense_candidates, _fetch_chunks_dense, _fetch_artifacts_dense and
_rrf_merge, with the same argument meaning and row shapes.  The SQL + pgvector scan is replaced by
a DenseTable: a DenseIndex in HBM plus the per-row columns the reference SELECTs, kept on the host.
Filters (_build_filter_clause, retrieve.py:93-120) become a row bitmask handed to the scan kernel.

The orchestrator (retrieve_evidence), the BM25 and tech-token lanes and the evidence-pack shaping
stay in the reference app; INTEGRATION.md shows the three call sites that switch to this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from datetime import datetime
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple, Union
from uuid import UUID

import numpy as np

from .config import settings
from .dense_index import DenseIndex

DEFAULT_RRF_K = 60
DEFAULT_DENSE_CHUNK_TOPK = 50
DEFAULT_DENSE_ARTIFACT_CHUNK_TOPK = 10


@dataclass
class RetrieveFilters:
    """Field-for-field the reference's pydantic RetrieveFilters (app/schemas.py:76-82)."""
    date_from: Optional[datetime] = None
    date_to: Optional[datetime] = None
    call_ids: Optional[List[UUID]] = None
    external_id: Optional[str] = None
    external_source: Optional[str] = None
    call_tags: Optional[List[str]] = None


def _resolve_call_ids(calls: Sequence[Dict[str, Any]], filters: Optional[RetrieveFilters]
                      ) -> Optional[List[UUID]]:
    """external_id (+ external_source) -> call ids, intersected with filters.call_ids
    (/root/reference/app/retrieve.py:46-90).  `calls` stands in for the `calls` table: dicts with
    call_id, external_id, external_source.  None means "no call scoping"; [] means "matches nothing"."""
    if not filters:
        return None
    call_ids: Optional[Set[UUID]] = set(filters.call_ids) if filters.call_ids else None
    if filters.external_id:
        resolved = {c["call_id"] for c in calls
                    if c.get("external_id") == filters.external_id
                    and (filters.external_source is None or c.get("external_source") == filters.external_source)}
        call_ids = (call_ids & resolved) if call_ids else resolved
    if call_ids is None:
        return None
    return sorted(call_ids, key=str)


def _rrf_merge(lanes: Dict[str, Sequence[Dict[str, Any]]], key_field: str, k: int = DEFAULT_RRF_K
               ) -> List[Tuple[Dict[str, Any], Set[str], float]]:
    """Reciprocal-rank fusion: score += 1/(k + rank), rank from 1; the first row seen for a key is
    kept; stable descending sort, so ties keep first-insertion order (lane order of the dict)."""
    score: Dict[Any, float] = {}
    first_row: Dict[Any, Dict[str, Any]] = {}
    hits: Dict[Any, Set[str]] = {}
    for lane, rows in lanes.items():
        for rank, row in enumerate(rows, start=1):
            key = row[key_field]
            score[key] = score.get(key, 0.0) + 1.0 / (k + rank)
            first_row.setdefault(key, row)
            hits.setdefault(key, set()).add(lane)
    order = sorted(score.items(), key=lambda kv: kv[1], reverse=True)
    return [(first_row[key], hits[key], s) for key, s in order]


def _vector_literal(values: Sequence[float]) -> str:
    return "[" + ",".join(format(float(v), ".10g") for v in values) + "]"


def _parse_vector(query_embedding: Union[str, Sequence[float], np.ndarray]) -> np.ndarray:
    if isinstance(query_embedding, str):
        body = query_embedding.strip()
        if not (body.startswith("[") and body.endswith("]")):
            raise ValueError("vector literal must look like '[v0,v1,...]'")
        return np.array([float(x) for x in body[1:-1].split(",")], dtype=np.float32)
    return np.asarray(query_embedding, dtype=np.float32)


def _dense_has_scoping(filters: Optional[RetrieveFilters], call_ids: Optional[Sequence[UUID]]) -> bool:
    if call_ids is not None:
        return True
    if not filters:
        return False
    return bool(filters.date_from or filters.date_to or filters.call_tags)


def _choose_dense_mode(estimated_rows: int, filters: Optional[RetrieveFilters],
                       call_ids: Optional[Sequence[UUID]]) -> str:
    """The reference's planner string.  The GPU lane always scans exactly; the function is kept so
    `notes.retrieval.dense_modes` keeps its meaning ("exact" / "ann" = what pgvector would do)."""
    if estimated_rows <= 0:
        return "exact"
    if _dense_has_scoping(filters, call_ids) and estimated_rows <= max(settings.embeddings_exact_scan_threshold, 0):
        return "exact"
    return "ann"


class DenseTable:
    """One embedded table (chunks or artifact_chunks): vectors in HBM + the SELECTed columns.

    columns: dict name -> sequence (one entry per row, same order as the vectors), must contain
    `id_field` and "call_id".  call_started_at: per-row datetime/np.datetime64 (the denormalised
    chunks.call_started_at column); call_tags: {call_id: [tags]} (calls.tags, joined on demand).
    """

    def __init__(self, name: str, id_field: str, *, dim: Optional[int] = None, capacity: int = 1 << 16,
                 device: Optional[int] = None) -> None:
        self.name = name
        self.id_field = id_field
        self.index = DenseIndex(dim or settings.embeddings_dim, capacity=capacity,
                                device=settings.embeddings_device if device is None else device)
        self.columns: Dict[str, list] = {}
        self.call_started_at = np.empty((0,), dtype="datetime64[us]")
        self.call_ids = np.empty((0,), dtype=object)
        self.call_tags: Dict[Any, Sequence[str]] = {}

    def __len__(self) -> int:
        return len(self.index)

    def close(self) -> None:
        self.index.close()

    def add(self, vectors, columns: Dict[str, Sequence[Any]], call_started_at: Optional[Sequence[Any]] = None,
            call_tags: Optional[Dict[Any, Sequence[str]]] = None) -> None:
        n = len(columns[self.id_field])
        if any(len(v) != n for v in columns.values()):
            raise ValueError("all columns must have one entry per vector")
        ids = np.asarray(columns[self.id_field], dtype=np.int64)
        self.index.add(vectors, ids=ids)
        for key, vals in columns.items():
            self.columns.setdefault(key, []).extend(list(vals))
        self.call_ids = np.concatenate([self.call_ids, np.asarray(list(columns["call_id"]), dtype=object)])
        ts = (np.asarray(list(call_started_at), dtype="datetime64[us]") if call_started_at is not None
              else np.full((n,), np.datetime64("NaT"), dtype="datetime64[us]"))
        self.call_started_at = np.concatenate([self.call_started_at, ts])
        if call_tags:
            self.call_tags.update(call_tags)
        self._pos_of_id = None

    # -- _build_filter_clause (retrieve.py:93-120) as a row mask --------------------------------
    def filter_mask(self, filters: Optional[RetrieveFilters], call_ids: Optional[Sequence[UUID]]
                    ) -> Optional[np.ndarray]:
        n = len(self)
        keep: Optional[np.ndarray] = None

        def land(cond: np.ndarray) -> None:
            nonlocal keep
            keep = cond if keep is None else (keep & cond)

        if filters:
            if filters.date_from:
                land(self.call_started_at >= np.datetime64(_naive_utc(filters.date_from), "us"))
            if filters.date_to:
                land(self.call_started_at <= np.datetime64(_naive_utc(filters.date_to), "us"))
            if call_ids is not None:
                wanted = set(call_ids)
                land(np.fromiter((c in wanted for c in self.call_ids), dtype=bool, count=n))
            if filters.call_tags:
                tags = set(filters.call_tags)
                ok_calls = {c for c, t in self.call_tags.items() if tags.intersection(t or ())}
                land(np.fromiter((c in ok_calls for c in self.call_ids), dtype=bool, count=n))
        return keep

    def estimate_candidates(self, filters: Optional[RetrieveFilters], call_ids: Optional[Sequence[UUID]]) -> int:
        """COUNT(*) ... WHERE <filters> AND embedding IS NOT NULL (retrieve.py:303-323)."""
        mask = self.filter_mask(filters, call_ids)
        return self.index.count_eligible(None if mask is None else DenseIndex.pack_mask(mask))

    def fetch_dense(self, query_embedding, filters: Optional[RetrieveFilters],
                    call_ids: Optional[Sequence[UUID]], mode: str, limit: int,
                    select: Sequence[str]) -> List[Dict[str, Any]]:
        """ORDER BY embedding <=> q LIMIT :limit, rows as mappings with `select` columns + score."""
        del mode  # the HBM scan is always exact; `mode` only labels what pgvector would have done
        if len(self) == 0 or limit <= 0:
            return []
        q = _parse_vector(query_embedding)
        mask = self.filter_mask(filters, call_ids)
        packed = None if mask is None else DenseIndex.pack_mask(mask)
        rows: List[Dict[str, Any]] = []
        ids, scores, counts = self.index.search(q[None, :], min(int(limit), _native_max_k()), row_mask=packed)
        pos_of = self._positions()
        for rid, sc in zip(ids[0, :counts[0]], scores[0, :counts[0]]):
            pos = pos_of[int(rid)]
            row = {name: self.columns[name][pos] for name in select}
            row["score"] = float(sc)
            rows.append(row)
        return rows

    def _positions(self) -> Dict[int, int]:
        if getattr(self, "_pos_of_id", None) is None:
            self._pos_of_id = {int(v): i for i, v in enumerate(self.columns[self.id_field])}
        return self._pos_of_id


def _native_max_k() -> int:
    from ._native import CRAG_MAX_K
    return CRAG_MAX_K


def _naive_utc(dt: datetime) -> datetime:
    if dt.tzinfo is not None:
        from datetime import timezone
        return dt.astimezone(timezone.utc).replace(tzinfo=None)
    return dt


CHUNK_SELECT = ("chunk_id", "call_id", "speaker", "start_ts_ms", "end_ts_ms", "text")
ARTIFACT_SELECT = ("artifact_chunk_id", "artifact_id", "call_id", "kind", "content")


def _estimate_dense_candidates(table: DenseTable, table_name: str, filters: Optional[RetrieveFilters],
                               call_ids: Optional[Sequence[UUID]]) -> int:
    del table_name
    return table.estimate_candidates(filters, call_ids)


def _fetch_chunks_dense(table: DenseTable, query_embedding, filters: Optional[RetrieveFilters],
                        call_ids: Optional[Sequence[UUID]], mode: str, limit: int) -> List[Dict[str, Any]]:
    """Same signature as the reference with the SQL connection replaced by the chunks DenseTable;
    rows: {chunk_id, call_id, speaker, start_ts_ms, end_ts_ms, text, score}, best first."""
    return table.fetch_dense(query_embedding, filters, call_ids, mode, limit, CHUNK_SELECT)


def _fetch_artifacts_dense(table: DenseTable, query_embedding, filters: Optional[RetrieveFilters],
                           call_ids: Optional[Sequence[UUID]], mode: str, limit: int) -> List[Dict[str, Any]]:
    """rows: {artifact_chunk_id, artifact_id, call_id, kind, content, score}, best first."""
    return table.fetch_dense(query_embedding, filters, call_ids, mode, limit, ARTIFACT_SELECT)
