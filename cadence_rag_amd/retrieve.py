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
        # bumped whenever rows are appended or move: whatever is built over row POSITIONS (the exact-token lane,
        # packed masks) is stale once it differs
        self.generation = 0
        # per-row tech_tokens (the `tech_tokens text[]` column), kept once a tech lane was built so that the lane
        # can be rebuilt when the table changes; None = not tracked
        self.tech_tokens: Optional[List[List[str]]] = None

    def __len__(self) -> int:
        return len(self.index)

    def close(self) -> None:
        self.index.close()

    def _take_tokens(self, columns: Dict[str, Sequence[Any]], n: int) -> Tuple[Dict[str, Sequence[Any]], List[List[str]]]:
        """Split an optional "tech_tokens" entry off the SELECTed columns (it feeds the exact-token lane, it is not
        a response column)."""
        if "tech_tokens" not in columns:
            return columns, [[] for _ in range(n)]
        columns = dict(columns)
        toks = [list(t or []) for t in columns.pop("tech_tokens")]
        if len(toks) != n:
            raise ValueError("tech_tokens must have one entry per vector")
        return columns, toks

    def add(self, vectors, columns: Dict[str, Sequence[Any]], call_started_at: Optional[Sequence[Any]] = None,
            call_tags: Optional[Dict[Any, Sequence[str]]] = None) -> None:
        """Append rows whose ids are ascending and above every stored id (the C ABI's contract); the
        index grows when its capacity is exhausted.  Rows that may arrive out of id order go through
        `insert`.  `columns` may carry "tech_tokens" (per-row token lists) for the exact-token lane."""
        n = len(columns[self.id_field])
        columns, toks = self._take_tokens(columns, n)
        if any(len(v) != n for v in columns.values()):
            raise ValueError("all columns must have one entry per vector")
        ids = np.asarray(columns[self.id_field], dtype=np.int64)
        self._reserve(n)
        self.index.add(vectors, ids=ids)
        for key, vals in columns.items():
            self.columns.setdefault(key, []).extend(list(vals))
        self.call_ids = np.concatenate([self.call_ids, np.asarray(list(columns["call_id"]), dtype=object)])
        ts = (np.asarray(list(call_started_at), dtype="datetime64[us]") if call_started_at is not None
              else np.full((n,), np.datetime64("NaT"), dtype="datetime64[us]"))
        self.call_started_at = np.concatenate([self.call_started_at, ts])
        if call_tags:
            self.call_tags.update(call_tags)
        if self.tech_tokens is not None:
            self.tech_tokens.extend(toks)
        self._pos_of_id = None
        self.generation += 1

    def _reserve(self, n_more: int) -> None:
        """Make room for n_more rows: the HBM index has a fixed capacity, so a full one is replaced by a
        larger one (x1.5) and the rows move device-to-device."""
        need = len(self.index) + int(n_more)
        if need <= self.index.capacity:
            return
        self._rebuild(capacity=max(need, int(self.index.capacity * 1.5) + 1024))

    def _rebuild(self, capacity: int, order: Optional[np.ndarray] = None, extra=None) -> None:
        """New index of `capacity` rows holding the current rows (+ `extra` = (vectors, ids)), permuted by
        `order` (positions into the concatenation) when given."""
        import torch
        old, n_old = self.index, len(self.index)
        dev = torch.device("cuda", old.device)
        new = DenseIndex(old.dim, capacity=capacity, device=old.device)
        try:
            if order is None and extra is None:
                step = 65536
                for lo in range(0, n_old, step):
                    m = min(step, n_old - lo)
                    buf = torch.empty(m, old.dim, dtype=torch.float32, device=dev)
                    ids = old.get_rows_into(lo, m, buf)
                    new.add(buf, ids=ids)
            else:
                rows = torch.empty(n_old, old.dim, dtype=torch.float32, device=dev)
                ids = old.get_rows_into(0, n_old, rows) if n_old else np.empty((0,), dtype=np.int64)
                if extra is not None:
                    rows = torch.cat([rows, _rows_on_device(extra[0], len(extra[1]), old.dim, dev)])
                    ids = np.concatenate([ids, np.asarray(extra[1], dtype=np.int64)])
                if order is not None:
                    rows = rows[torch.as_tensor(order, device=dev)]
                    ids = ids[order]
                new.add(rows, ids=ids)
        except Exception:
            new.close()
            raise
        self.index = new
        old.close()
        self.generation += 1

    def insert(self, vectors, columns: Dict[str, Sequence[Any]], call_started_at: Optional[Sequence[Any]] = None,
               call_tags: Optional[Dict[Any, Sequence[str]]] = None) -> None:
        """`add` for rows in any id order (a row embedded late has an id below the stored maximum): when the
        new ids do not simply continue the stored ones, the table is rebuilt in ascending id order, which
        is the order every tie-break of the lane assumes.  An id that is already stored is an error (use
        DenseIndex.update to re-embed in place)."""
        new_ids = np.asarray(columns[self.id_field], dtype=np.int64)
        n = int(new_ids.size)
        if n == 0:
            return
        old_ids = np.asarray(self.columns.get(self.id_field, []), dtype=np.int64)
        if (old_ids.size == 0 or new_ids.min() > old_ids[-1]) and np.all(np.diff(new_ids) > 0):
            return self.add(vectors, columns, call_started_at, call_tags)
        columns, toks = self._take_tokens(columns, n)
        if any(len(v) != n for v in columns.values()):
            raise ValueError("all columns must have one entry per vector")
        all_ids = np.concatenate([old_ids, new_ids])
        order = np.argsort(all_ids, kind="stable")
        if np.any(np.diff(all_ids[order]) == 0):
            raise ValueError(f"duplicate {self.id_field} in insert")
        # `vectors` may be a CUDA tensor (the device-resident backfill): it stays on the device
        self._rebuild(capacity=max(self.index.capacity, all_ids.size), order=order, extra=(vectors, new_ids))
        for key in set(self.columns) | set(columns):
            merged = list(self.columns.get(key, [None] * old_ids.size)) + list(columns.get(key, [None] * n))
            self.columns[key] = [merged[i] for i in order]
        ts = (np.asarray(list(call_started_at), dtype="datetime64[us]") if call_started_at is not None
              else np.full((n,), np.datetime64("NaT"), dtype="datetime64[us]"))
        self.call_started_at = np.concatenate([self.call_started_at, ts])[order]
        self.call_ids = np.concatenate([self.call_ids, np.asarray(list(columns["call_id"]), dtype=object)])[order]
        if call_tags:
            self.call_tags.update(call_tags)
        if self.tech_tokens is not None:
            merged_t = self.tech_tokens + toks
            self.tech_tokens = [merged_t[i] for i in order]
        self._pos_of_id = None
        self.generation += 1

    def sink(self, row_columns):
        """Backfill sink (embedding_pipeline.BackfillStore.update_embeddings -> HBM): an object whose
        `add(vectors, ids=...)` asks `row_columns(ids)` for the rows' SELECTed columns — a dict of column
        lists that may also carry "call_started_at" and "tech_tokens" — and inserts them here, so that the
        host-side columns, the filter masks, the exact-token lane and the index stay one table.  `vectors` may be
        host lists / arrays or a CUDA tensor (DeviceSinkStore)."""
        table = self

        class _Sink:
            def add(self, vectors, ids):
                cols = dict(row_columns(list(ids)))
                started = cols.pop("call_started_at", None)
                cols.setdefault(table.id_field, list(ids))
                table.insert(vectors, cols, call_started_at=started)

        return _Sink()

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

    @classmethod
    def from_rows(cls, name: str, id_field: str, rows, *, select: Sequence[str], dim: Optional[int] = None,
                  call_tags: Optional[Dict[Any, Sequence[str]]] = None, batch: int = 65536,
                  device: Optional[int] = None, headroom: float = 0.25) -> Tuple["DenseTable", List[List[str]]]:
        """Startup loader: `rows` are the mappings of
            SELECT <select>, call_started_at, tech_tokens, embedding FROM <name>
            WHERE embedding IS NOT NULL ORDER BY <id_field>
        with `embedding` in any of pgvector's forms (text literal '[v,...]', binary send/recv bytes, or a
        sequence of floats).  Returns the table and the per-row tech_tokens (for build_tech_lane).  Rows must
        come in ascending id order so that equal scores resolve to the lower id, as ORDER BY does.  The
        index is allocated with `headroom` spare capacity for the rows ingest / backfill add later (it
        also grows on demand, see `_reserve`)."""
        from . import vector_io
        rows = list(rows)
        d = dim or settings.embeddings_dim
        table = cls(name, id_field, dim=d, capacity=max(int(len(rows) * (1.0 + max(headroom, 0.0))) + 64, 1),
                    device=device)
        tokens: List[List[str]] = []
        last_id = None
        for lo in range(0, len(rows), batch):
            part = rows[lo:lo + batch]
            vecs = np.empty((len(part), d), dtype=np.float32)
            for i, row in enumerate(part):
                emb = row["embedding"]
                if isinstance(emb, str):
                    vecs[i] = vector_io.parse_vector(emb, d)
                elif isinstance(emb, (bytes, bytearray, memoryview)):
                    v = vector_io.from_binary(bytes(emb))
                    if v.size != d:
                        raise ValueError(f"expected {d} dimensions, not {v.size}")
                    vecs[i] = v
                else:
                    vecs[i] = np.asarray(emb, dtype=np.float32).reshape(d)
                rid = row[id_field]
                if last_id is not None and rid <= last_id:
                    raise ValueError(f"rows must be in ascending {id_field} order ({rid} after {last_id})")
                last_id = rid
                tokens.append(list(row.get("tech_tokens") or []))
            cols = {c: [row[c] for row in part] for c in select}
            table.add(vecs, cols, call_started_at=[row.get("call_started_at") for row in part],
                      call_tags=call_tags if lo == 0 else None)
        return table, tokens

    def build_tech_lane(self, row_tokens: Optional[Sequence[Sequence[str]]] = None):
        """GPU exact-token lane over this table's rows (row i <-> position i, so filter masks are shared):
        the `tech_tokens text[]` column + ORDER BY call_started_at DESC, id ASC (retrieve.py:183-242).
        The table keeps the tokens from here on (rows added later bring theirs in `columns["tech_tokens"]`), and
        the lane remembers the table generation it was built for: GpuRetrieveBackend rebuilds a stale lane."""
        import torch

        from .fusion import TechTokenIndex
        if row_tokens is None:
            row_tokens = self.tech_tokens
            if row_tokens is None:
                raise ValueError("this table does not track tech_tokens yet: pass row_tokens")
        if len(row_tokens) != len(self):
            raise ValueError("row_tokens must have one entry per table row")
        self.tech_tokens = [list(t or []) for t in row_tokens]
        lane = TechTokenIndex(self.tech_tokens, np.asarray(self.columns[self.id_field], dtype=np.int64),
                              self.call_started_at, torch.device("cuda", self.index.device))
        lane.table_generation = self.generation
        return lane

    def _positions(self) -> Dict[int, int]:
        if getattr(self, "_pos_of_id", None) is None:
            self._pos_of_id = {int(v): i for i, v in enumerate(self.columns[self.id_field])}
        return self._pos_of_id


def _rows_on_device(vectors, n: int, dim: int, dev):
    """[n, dim] float32 on `dev` from host lists / numpy or from a torch tensor (a CUDA tensor never visits the
    host)."""
    import torch
    if isinstance(vectors, torch.Tensor):
        return vectors.to(device=dev, dtype=torch.float32).reshape(n, dim)
    return torch.as_tensor(np.asarray(vectors, dtype=np.float32).reshape(n, dim), device=dev)


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


# ------------------------------------------------------------------------------------------------
# /retrieve entry point (S9): the orchestration of /root/reference/app/retrieve.py:392-688 with the
# SQL connection replaced by a backend object.  Lane order, RRF, ids_only ordering, evidence packing,
# budget clipping and every `notes.retrieval` / `debug` key follow the reference; goldens captured
# from the reference's own retrieve_evidence (tests/golden/reference_retrieve_evidence.json) pin it.
# ------------------------------------------------------------------------------------------------
DEFAULT_CHUNK_BM25_TOPK = 50
DEFAULT_ARTIFACT_CHUNK_BM25_TOPK = 10
DEFAULT_TECH_TOPK = 50
DEFAULT_MAX_ARTIFACTS = 2
DEFAULT_MAX_QUOTES_PER_CALL = 2
DEFAULT_SNIPPET_CHARS = 800


@dataclass
class Budget:
    """app/schemas.py:71-73."""
    max_evidence_items: int = 8
    max_total_chars: int = 6000

    def model_dump(self) -> Dict[str, int]:
        return {"max_evidence_items": self.max_evidence_items, "max_total_chars": self.max_total_chars}


@dataclass
class RetrieveRequest:
    """app/schemas.py:85-93."""
    query: str
    intent: str = "auto"
    filters: Optional[RetrieveFilters] = None
    budget: Optional[Budget] = None
    return_style: str = "evidence_pack_json"
    debug: bool = False


def _clip(text: str, max_chars: int) -> str:
    """retrieve.py:24-29."""
    if max_chars <= 0:
        return ""
    if len(text) <= max_chars:
        return text
    return text[: max_chars - 1].rstrip() + "…"


def _build_debug_lane(rows: Sequence[Dict[str, Any]], id_field: str) -> List[Dict[str, Any]]:
    """retrieve.py:32-41."""
    return [{id_field: row[id_field], "rank": rank, "score": row.get("score")}
            for rank, row in enumerate(rows, start=1)]


class RetrieveBackend:
    """What retrieve_evidence needs from storage: one method per SQL helper of the reference
    (retrieve.py:46-389), same arguments minus the connection.  Subclass or duck-type."""

    def resolve_call_ids(self, filters): return None
    def fetch_chunks_bm25(self, query, filters, call_ids, limit): return []
    def fetch_artifacts_bm25(self, query, filters, call_ids, limit): return []
    def fetch_chunks_tech(self, tokens, filters, call_ids, limit): return []
    def fetch_artifacts_tech(self, tokens, filters, call_ids, limit): return []
    def estimate_dense_candidates(self, table_name, filters, call_ids): return 0
    def fetch_chunks_dense(self, query_embedding, filters, call_ids, mode, limit): return []
    def fetch_artifacts_dense(self, query_embedding, filters, call_ids, mode, limit): return []


class GpuRetrieveBackend(RetrieveBackend):
    """Dense lanes from the HBM-resident DenseTables, exact-token lanes from GPU TechTokenIndex objects
    (cadence_rag_amd.fusion) when attached, BM25 lanes from injected callables (pg_search's arithmetic
    is not in the reference repository: its rows are an input here, as they are to _rrf_merge)."""

    def __init__(self, chunks: DenseTable, artifact_chunks: DenseTable, *, calls: Sequence[Dict[str, Any]] = (),
                 bm25_chunks=None, bm25_artifacts=None, tech_chunks=None, tech_artifacts=None) -> None:
        self.tables = {"chunks": chunks, "artifact_chunks": artifact_chunks}
        self.calls = list(calls)
        self._bm25 = {"chunks": bm25_chunks, "artifact_chunks": bm25_artifacts}
        self._tech = {"chunks": tech_chunks, "artifact_chunks": tech_artifacts}

    def resolve_call_ids(self, filters):
        return _resolve_call_ids(self.calls, filters)

    def fetch_chunks_bm25(self, query, filters, call_ids, limit):
        fn = self._bm25["chunks"]
        return list(fn(query, filters, call_ids, limit)) if fn else []

    def fetch_artifacts_bm25(self, query, filters, call_ids, limit):
        fn = self._bm25["artifact_chunks"]
        return list(fn(query, filters, call_ids, limit)) if fn else []

    def _tech_rows(self, name, select, tokens, filters, call_ids, limit):
        lane, table = self._tech[name], self.tables[name]
        if not tokens or lane is None or len(table) == 0:
            return []
        import torch
        if getattr(lane, "table_generation", table.generation) != table.generation or lane.n != len(table):
            # rows were appended or moved since the lane was built: its row positions (and with them every packed
            # mask bit) no longer mean the table's rows
            if table.tech_tokens is None or len(table.tech_tokens) != len(table):
                raise RuntimeError(f"the exact-token lane of {name} is stale (table generation {table.generation}) "
                                   "and the table does not track tech_tokens: rebuild it with build_tech_lane")
            lane = self._tech[name] = table.build_tech_lane()
        mask = table.filter_mask(filters, call_ids)
        d_mask = None
        if mask is not None:
            d_mask = torch.from_numpy(DenseIndex.pack_mask(mask)).to(lane.device)
        ids, counts = lane.search([list(tokens)], int(limit), row_mask=d_mask, mask_stride=0)
        pos_of = table._positions()
        out = []
        for rid in ids[0, : int(counts[0])].tolist():
            pos = pos_of[int(rid)]
            out.append({col: table.columns[col][pos] for col in select})
        return out

    def fetch_chunks_tech(self, tokens, filters, call_ids, limit):
        return self._tech_rows("chunks", CHUNK_SELECT, tokens, filters, call_ids, limit)

    def fetch_artifacts_tech(self, tokens, filters, call_ids, limit):
        return self._tech_rows("artifact_chunks", ARTIFACT_SELECT, tokens, filters, call_ids, limit)

    def estimate_dense_candidates(self, table_name, filters, call_ids):
        return _estimate_dense_candidates(self.tables[table_name], table_name, filters, call_ids)

    def fetch_chunks_dense(self, query_embedding, filters, call_ids, mode, limit):
        return _fetch_chunks_dense(self.tables["chunks"], query_embedding, filters, call_ids, mode, limit)

    def fetch_artifacts_dense(self, query_embedding, filters, call_ids, mode, limit):
        return _fetch_artifacts_dense(self.tables["artifact_chunks"], query_embedding, filters, call_ids, mode, limit)


_backend: Optional[RetrieveBackend] = None


def set_backend(backend: Optional[RetrieveBackend]) -> None:
    """Register the process-wide backend (the counterpart of the reference's module-level `engine`)."""
    global _backend
    _backend = backend


@dataclass(frozen=True)
class _Side:
    """One of the two evidence tables as /retrieve sees it: which lanes feed it, how one of its rows becomes
    an evidence item and which caps apply.  The response of /root/reference/app/retrieve.py:392-688 is a
    function of these two records and the request; tests/golden/reference_retrieve_evidence.json pins it."""
    table: str          # backend table name, also the key inside notes/debug dictionaries
    out: str            # response list ("artifacts" / "quotes") and debug.lanes key
    debug_key: str
    id_field: str
    tag: str            # ids_only prefix
    rank: int           # ids_only tie order: lower first
    letter: str         # evidence_id prefix
    body: str           # column the snippet is cut from
    carry: Tuple[str, ...]   # columns copied into the item, in response order
    bm25_topk: int
    dense_topk: int
    list_cap: Optional[int]  # at most this many items of this kind
    per_call: Optional[int]  # at most this many per call_id


_SIDES = (
    _Side("artifact_chunks", "artifacts", "artifacts", "artifact_chunk_id", "artifact_chunk", 0, "A", "content",
          ("artifact_id", "artifact_chunk_id", "kind"), DEFAULT_ARTIFACT_CHUNK_BM25_TOPK,
          DEFAULT_DENSE_ARTIFACT_CHUNK_TOPK, DEFAULT_MAX_ARTIFACTS, None),
    _Side("chunks", "quotes", "chunks", "chunk_id", "chunk", 1, "Q", "text",
          ("chunk_id", "speaker", "start_ts_ms", "end_ts_ms"), DEFAULT_CHUNK_BM25_TOPK,
          DEFAULT_DENSE_CHUNK_TOPK, None, DEFAULT_MAX_QUOTES_PER_CALL),
)
_BY_TABLE = {s.table: s for s in _SIDES}
# the backend is queried chunks first (the reference's statement order; the replay backend records it)
_QUERY_ORDER = (_BY_TABLE["chunks"], _BY_TABLE["artifact_chunks"])


class _Purse:
    """The request budget while the evidence pack is being filled."""

    def __init__(self, budget: Budget) -> None:
        self.items = budget.max_evidence_items
        self.chars = budget.max_total_chars

    @property
    def spent(self) -> bool:
        return self.items <= 0 or self.chars <= 0


def _pack(ranked, side: _Side, purse: _Purse, item_cap: Optional[int]) -> List[Dict[str, Any]]:
    """Ranked rows of one side -> evidence items, best first, until the purse or the side's caps run out.
    A row over its call's quota is passed over without ending the walk."""
    out: List[Dict[str, Any]] = []
    used_by_call: Dict[str, int] = {}
    for row, lanes, _ in ranked:
        if purse.spent or (item_cap is not None and len(out) >= item_cap):
            break
        call = str(row["call_id"])
        if side.per_call is not None:
            if used_by_call.get(call, 0) >= side.per_call:
                continue
            used_by_call[call] = used_by_call.get(call, 0) + 1
        snippet = _clip(row[side.body], min(DEFAULT_SNIPPET_CHARS, purse.chars))
        purse.chars -= len(snippet)
        purse.items -= 1
        item = {"evidence_id": f"{side.letter}-{row[side.id_field]}", "call_id": call}
        item.update((col, row[col]) for col in side.carry)
        item["snippet"] = snippet
        item["why_relevant"] = " + ".join(sorted(lanes))
        out.append(item)
    return out


class _DenseState:
    """What the dense lane contributed to one request (all of it is echoed in notes / debug)."""

    def __init__(self) -> None:
        self.on = False
        self.model_id: Optional[str] = None
        self.error: Optional[str] = None
        self.literal: Optional[str] = None
        self.mode: Dict[str, Optional[str]] = {s.table: None for s in _QUERY_ORDER}
        self.candidates: Dict[str, int] = {s.table: 0 for s in _QUERY_ORDER}

    def embed(self, query: str) -> None:
        from . import embeddings as _emb
        self.on = _emb.embeddings_enabled()
        if not self.on:
            return
        try:  # fail-open: a broken encoder turns the lane off for this request and is reported
            res = _emb.embed_texts([query])
        except _emb.EmbeddingClientError as exc:
            self.on, self.error = False, str(exc)
            return
        self.model_id, self.literal = res.model, _vector_literal(res.vectors[0])

    @property
    def planner(self) -> str:
        if not self.on:
            return "lexical_only"
        return "ann" if "ann" in self.mode.values() else "exact"

    def topk(self, side: _Side) -> int:
        return side.dense_topk if self.on else 0


def _gather_lanes(be: "RetrieveBackend", query: str, tokens: List[str], filters, dense: _DenseState
                  ) -> Dict[str, Dict[str, Sequence[Dict[str, Any]]]]:
    """table -> {lane name -> rows}, lanes in fusion order (bm25, tech_tokens, dense)."""
    call_ids = be.resolve_call_ids(filters)
    fetch = {"chunks": (be.fetch_chunks_bm25, be.fetch_chunks_tech, be.fetch_chunks_dense),
             "artifact_chunks": (be.fetch_artifacts_bm25, be.fetch_artifacts_tech, be.fetch_artifacts_dense)}
    lanes: Dict[str, Dict[str, Sequence[Dict[str, Any]]]] = {s.table: {} for s in _QUERY_ORDER}
    for s in _QUERY_ORDER:
        lanes[s.table]["bm25"] = fetch[s.table][0](query, filters, call_ids, s.bm25_topk)
    for s in _QUERY_ORDER:
        lanes[s.table]["tech_tokens"] = fetch[s.table][1](tokens, filters, call_ids, DEFAULT_TECH_TOPK)
    if dense.on and dense.literal is not None:
        for s in _QUERY_ORDER:
            dense.candidates[s.table] = be.estimate_dense_candidates(s.table, filters, call_ids)
        for s in _QUERY_ORDER:
            dense.mode[s.table] = _choose_dense_mode(dense.candidates[s.table], filters, call_ids)
    if dense.on:
        for s in _QUERY_ORDER:
            lanes[s.table]["dense"] = (fetch[s.table][2](dense.literal, filters, call_ids, dense.mode[s.table],
                                                         s.dense_topk) if dense.literal is not None else [])
    return lanes


def _debug_section(lanes, dense: _DenseState) -> Dict[str, Any]:
    chunks, artifacts = _BY_TABLE["chunks"], _BY_TABLE["artifact_chunks"]
    return {
        "lanes": {s.debug_key: {name: _build_debug_lane(rows, s.id_field) for name, rows in lanes[s.table].items()}
                  for s in _QUERY_ORDER},
        "limits": {"bm25_chunk_topk": chunks.bm25_topk, "bm25_artifact_chunk_topk": artifacts.bm25_topk,
                   "tech_token_topk": DEFAULT_TECH_TOPK, "dense_chunk_topk": dense.topk(chunks),
                   "dense_artifact_chunk_topk": dense.topk(artifacts)},
        "dense": {"enabled": dense.on, "model_id": dense.model_id, "error": dense.error,
                  "modes": dict(dense.mode), "candidate_rows": dict(dense.candidates)},
    }


def _retrieval_notes(tokens: List[str], dense: _DenseState) -> Dict[str, Any]:
    chunks, artifacts = _BY_TABLE["chunks"], _BY_TABLE["artifact_chunks"]
    return {
        "planner": dense.planner,
        "dense_topk": max(dense.topk(chunks), dense.topk(artifacts)),
        "lex_topk": chunks.bm25_topk,
        "artifact_chunk_lex_topk": artifacts.bm25_topk,
        "reranked_from": None,
        "bm25_chunk_topk": chunks.bm25_topk,
        "bm25_artifact_chunk_topk": artifacts.bm25_topk,
        "tech_token_topk": DEFAULT_TECH_TOPK,
        "tech_tokens": tokens,
        "lanes": {"bm25": True, "tech_tokens": True, "dense": dense.on},
        "dense_model_id": dense.model_id,
        "dense_error": dense.error,
        "dense_modes": dict(dense.mode),
        "dense_candidate_rows": dict(dense.candidates),
        "hnsw_ef_search": settings.embeddings_hnsw_ef_search if dense.on else None,
    }


def retrieve_evidence(payload: RetrieveRequest, backend: Optional[RetrieveBackend] = None) -> Dict[str, Any]:
    """The /retrieve entry point (S9; reference: /root/reference/app/retrieve.py:392-688): lexical lanes and
    the dense lane per table -> RRF -> either the fused id list or a budgeted evidence pack.  Built from the
    response contract (ten reference-captured scenarios), table-driven over `_SIDES`."""
    from uuid import uuid4

    from .tech_tokens import extract_tech_tokens

    be = backend if backend is not None else _backend
    if be is None:
        raise RuntimeError("retrieve_evidence: no backend registered (set_backend)")
    head: Dict[str, Any] = {"query_id": str(uuid4())}
    ids_only = payload.return_style == "ids_only"
    budget = payload.budget or Budget()
    query = payload.query.strip()
    if not query:
        if ids_only:
            return {**head, "retrieved_ids": []}
        return {**head, "intent": payload.intent, "budget": budget.model_dump(),
                **{s.out: [] for s in _SIDES}, "notes": {"error": "empty query"}}

    tokens = extract_tech_tokens(query)
    dense = _DenseState()
    dense.embed(query)
    lanes = _gather_lanes(be, query, tokens, payload.filters, dense)
    fused = {s.table: _rrf_merge(lanes[s.table], s.id_field) for s in _SIDES}

    if ids_only:
        flat = [(-score, s.rank, row[s.id_field], s.tag) for s in _SIDES for row, _, score in fused[s.table]]
        flat.sort(key=lambda t: t[:3])
        response = {**head, "retrieved_ids": [f"{tag}:{rid}" for _, _, rid, tag in flat]}
    else:
        purse = _Purse(budget)
        packed = {}
        for s in _SIDES:  # artifacts are packed first and share the purse with the quotes
            cap = None if s.list_cap is None else min(s.list_cap, budget.max_evidence_items)
            packed[s.out] = _pack(fused[s.table], s, purse, cap)
        response = {**head, "intent": payload.intent, "budget": budget.model_dump(), **packed,
                    "notes": {"retrieval": _retrieval_notes(tokens, dense)}}
    if payload.debug:
        response["debug"] = _debug_section(lanes, dense)
    return response
