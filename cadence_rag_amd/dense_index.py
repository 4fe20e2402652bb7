"""DenseIndex — the in-HBM replacement for the reference's `embedding vector(1024)` column and
its pgvector scan (reference: /root/reference/app/retrieve.py:326-389,
alembic/versions/0001_initial_schema.py:87).  Thin Python over the C ABI (include/crag_dense.h);
all arithmetic happens in the HIP library.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _native

try:  # torch is plumbing only (device buffers / streams); numpy-only use works without it
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _as_f32_2d(x, dim: int, what: str):
    """Return (pointer, rows, keepalive) for a [n, dim] float32 C-contiguous numpy/torch array."""
    if _is_torch(x):
        t = x
        if t.dtype != torch.float32:
            t = t.to(torch.float32)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        t = t.contiguous()
        if t.dim() != 2 or t.shape[1] != dim:
            raise ValueError(f"{what} must have shape [n, {dim}], got {tuple(t.shape)}")
        return t.data_ptr(), int(t.shape[0]), t
    a = np.asarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    a = np.ascontiguousarray(a)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"{what} must have shape [n, {dim}], got {a.shape}")
    return a.ctypes.data, int(a.shape[0]), a


class DenseIndex:
    """Exact cosine top-k over fp32 rows resident in one GPU's HBM."""

    def __init__(self, dim: int = 1024, capacity: int = 1 << 20, device: int = 0) -> None:
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        self.dim = int(dim)
        self.device = int(device)
        _native.check(self._lib.crag_index_create(self.device, self.dim, int(capacity),
                                                  ctypes.byref(self._h)), "crag_index_create")

    # -- lifecycle -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.crag_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "DenseIndex":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __len__(self) -> int:
        return int(self._lib.crag_index_size(self._h))

    @property
    def capacity(self) -> int:
        return int(self._lib.crag_index_capacity(self._h))

    # -- corpus ----------------------------------------------------------------------------
    def add(self, vectors, ids=None) -> None:
        ptr, n, keep = _as_f32_2d(vectors, self.dim, "vectors")
        ids_ptr, keep_ids = None, None
        if ids is not None:
            if _is_torch(ids):
                keep_ids = ids.to(torch.int64).contiguous()
                if keep_ids.numel() != n:
                    raise ValueError("ids length mismatch")
                ids_ptr = keep_ids.data_ptr()
            else:
                keep_ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
                if keep_ids.size != n:
                    raise ValueError("ids length mismatch")
                ids_ptr = keep_ids.ctypes.data
        _native.check(self._lib.crag_index_add(self._h, ptr, ids_ptr, n), "crag_index_add")
        del keep, keep_ids

    def update(self, pos: int, vectors) -> None:
        ptr, n, keep = _as_f32_2d(vectors, self.dim, "vectors")
        _native.check(self._lib.crag_index_update(self._h, int(pos), ptr, n), "crag_index_update")
        del keep

    def get_rows(self, pos: int, n: int) -> Tuple[np.ndarray, np.ndarray]:
        rows = np.empty((n, self.dim), dtype=np.float32)
        ids = np.empty((n,), dtype=np.int64)
        _native.check(self._lib.crag_index_get_rows(self._h, int(pos), int(n), rows.ctypes.data,
                                                    ids.ctypes.data), "crag_index_get_rows")
        return rows, ids

    def get_rows_into(self, pos: int, n: int, d_rows) -> np.ndarray:
        """Device-to-device form of get_rows: rows [pos, pos+n) into the float32 CUDA tensor d_rows
        ([n, dim], contiguous); returns their ids (host)."""
        if tuple(d_rows.shape) != (n, self.dim) or not d_rows.is_contiguous() or d_rows.dtype != torch.float32:
            raise ValueError(f"d_rows must be a contiguous float32 [{n}, {self.dim}] tensor")
        ids = np.empty((n,), dtype=np.int64)
        if n:
            _native.check(self._lib.crag_index_get_rows(self._h, int(pos), int(n), d_rows.data_ptr(),
                                                        ids.ctypes.data), "crag_index_get_rows")
        return ids

    @staticmethod
    def pack_mask(eligible) -> np.ndarray:
        """bool [n] or [nq, n] -> the bit mask the C ABI takes (bit i&7 of byte i>>3), with each
        row padded to a multiple of 4 bytes."""
        e = np.asarray(eligible, dtype=bool)
        packed = np.packbits(e, axis=-1, bitorder="little")
        pad = (-packed.shape[-1]) % 4
        if pad:
            width = [(0, 0)] * (packed.ndim - 1) + [(0, pad)]
            packed = np.pad(packed, width)
        return np.ascontiguousarray(packed)

    def count_eligible(self, row_mask: Optional[np.ndarray] = None) -> int:
        out = ctypes.c_int64(0)
        ptr = None
        if row_mask is not None:
            row_mask = np.ascontiguousarray(np.asarray(row_mask, dtype=np.uint8))
            ptr = row_mask.ctypes.data
        _native.check(self._lib.crag_index_count_eligible(self._h, ptr, ctypes.byref(out)),
                      "crag_index_count_eligible")
        return int(out.value)

    # -- search ----------------------------------------------------------------------------
    def search(self, queries, k: int, row_mask: Optional[np.ndarray] = None
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Synchronous exact top-k.  Returns (ids [nq,k] int64 (-1 pad), scores [nq,k] float32
        (NaN pad), counts [nq] int32).  row_mask: packed bits from pack_mask(), shape
        [bytes] (shared) or [nq, bytes] (per query)."""
        ptr, nq, keep = _as_f32_2d(queries, self.dim, "queries")
        if not 1 <= int(k) <= _native.CRAG_MAX_K:
            raise ValueError(f"k must be in [1, {_native.CRAG_MAX_K}] (got {k})")
        out_ids = np.empty((nq, k), dtype=np.int64)
        out_scores = np.empty((nq, k), dtype=np.float32)
        out_counts = np.empty((nq,), dtype=np.int32)
        mptr, stride = None, 0
        if row_mask is not None:
            row_mask = np.ascontiguousarray(np.asarray(row_mask, dtype=np.uint8))
            need = ((len(self) + 31) // 32) * 4
            if row_mask.shape[-1] < need:
                raise ValueError(f"row_mask needs {need} bytes per row (use pack_mask)")
            if row_mask.ndim == 2:
                if row_mask.shape[0] != nq:
                    raise ValueError("per-query row_mask must have one row per query")
                stride = row_mask.shape[1]
                if stride % 4:
                    raise ValueError("row_mask row stride must be a multiple of 4 bytes")
            mptr = row_mask.ctypes.data
        _native.check(self._lib.crag_index_search(self._h, ptr, nq, int(k), mptr, stride,
                                                  out_ids.ctypes.data, out_scores.ctypes.data,
                                                  out_counts.ctypes.data), "crag_index_search")
        del keep
        return out_ids, out_scores, out_counts

    def search_async(self, d_queries, k: int, d_out_ids, d_out_scores, d_out_counts,
                     d_row_mask=None, mask_stride: int = 0, stream: int = 0) -> None:
        """All-device search enqueued on `stream` (a hipStream_t as int, e.g.
        torch.cuda.current_stream().cuda_stream).  Arguments are torch CUDA tensors."""
        nq = int(d_queries.shape[0])
        _native.check(self._lib.crag_index_search_async(
            self._h, d_queries.data_ptr(), nq, int(k),
            None if d_row_mask is None else d_row_mask.data_ptr(), int(mask_stride),
            d_out_ids.data_ptr(), d_out_scores.data_ptr(), d_out_counts.data_ptr(),
            ctypes.c_void_p(stream)), "crag_index_search_async")

    def search_pipelined(self, d_queries, k: int, d_out_ids, d_out_scores, d_out_counts,
                         d_row_mask=None, mask_stride: int = 0, stream: int = 0, inputs_ready: bool = False) -> None:
        """Throughput form for a run of INDEPENDENT searches issued from one stream (crag_index_search_pipelined):
        consecutive calls rotate over three streams of the index's own (k <= 24; larger k runs in order), so search
        i + 1's preparation and scan run beside search i's selection.  The outputs are defined on `stream` only behind join(stream).  inputs_ready: the
        queries / mask are complete in memory now (no event orders the internal stream behind `stream`)."""
        nq = int(d_queries.shape[0])
        _native.check(self._lib.crag_index_search_pipelined(
            self._h, d_queries.data_ptr(), nq, int(k),
            None if d_row_mask is None else d_row_mask.data_ptr(), int(mask_stride),
            d_out_ids.data_ptr(), d_out_scores.data_ptr(), d_out_counts.data_ptr(),
            ctypes.c_void_p(stream), 1 if inputs_ready else 0), "crag_index_search_pipelined")

    def join(self, stream: int = 0) -> None:
        """Make `stream` wait for every pipelined search issued so far (does not block the host)."""
        _native.check(self._lib.crag_index_join(self._h, ctypes.c_void_p(stream)), "crag_index_join")

    # -- profiling / reporting -------------------------------------------------------------
    def profile_enable(self, every: int = 1) -> None:
        """Record HIP events around the scan/merge kernels of every `every`-th search (0 = off)."""
        _native.check(self._lib.crag_index_profile_enable(self._h, int(every)), "profile_enable")

    def profile_read(self) -> Tuple[int, float, float]:
        n = ctypes.c_int64(0)
        scan = ctypes.c_double(0.0)
        merge = ctypes.c_double(0.0)
        _native.check(self._lib.crag_index_profile_read(self._h, ctypes.byref(n), ctypes.byref(scan),
                                                        ctypes.byref(merge)), "profile_read")
        return int(n.value), float(scan.value), float(merge.value)

    def profile_read_ex(self) -> Tuple[int, float, float, float]:
        """(samples, scan ms total, rest ms total, back-to-back event pair ms total): see crag_index_profile_read_ex."""
        n = ctypes.c_int64(0)
        scan, merge, pair = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        _native.check(self._lib.crag_index_profile_read_ex(self._h, ctypes.byref(n), ctypes.byref(scan),
                                                           ctypes.byref(merge), ctypes.byref(pair)), "profile_read_ex")
        return int(n.value), float(scan.value), float(merge.value), float(pair.value)

    def prefilter_stats(self) -> dict:
        """Searches / candidates / exactly rescored rows of the prefilter path since the last call."""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _native.check(self._lib.crag_index_prefilter_stats(self._h, ctypes.byref(a), ctypes.byref(b),
                                                           ctypes.byref(c)), "prefilter_stats")
        return {"searches": a.value, "candidates": b.value, "rescored_rows": c.value}

    def phase_trace(self):
        """Developer probe (CRAG_PHASE_TRACE=1 when the index was created): see crag_index_phase_trace."""
        buf = (ctypes.c_uint64 * 128)()
        _native.check(self._lib.crag_index_phase_trace(self._h, buf), "phase_trace")
        return [int(v) for v in buf]

    def prefilter_row_bytes(self) -> int:
        """Bytes of one corpus row the prefilter scan streams (2 KiB with the fp16 mirror, 4 KiB without, 0 = off)."""
        return int(self._lib.crag_index_prefilter_row_bytes(self._h))

    def last_scan_kernel(self) -> str:
        """Name of the scan kernel the most recent search launched (as rocprofv3 prints it)."""
        name = self._lib.crag_index_last_scan_kernel(self._h)
        return name.decode() if name else ""

    def scan_geometry(self, nq: int) -> dict:
        wg, th, qb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        ab = ctypes.c_int64(0)
        _native.check(self._lib.crag_index_scan_geometry(self._h, int(nq), ctypes.byref(wg),
                                                         ctypes.byref(th), ctypes.byref(qb),
                                                         ctypes.byref(ab)), "scan_geometry")
        return {"workgroups": wg.value, "threads": th.value, "query_blocks": qb.value,
                "algorithmic_bytes": int(ab.value)}


def merge_topk(d_ids, d_scores, d_counts, d_out_ids, d_out_scores, d_out_counts, stream: int = 0,
               device: Optional[int] = None) -> None:
    """Merge [n_lists, nq, k] per-shard results (torch CUDA tensors) into [nq, k] on the GPU."""
    lib = _native.load()
    n_lists, nq, k = (int(v) for v in d_ids.shape)
    dev = d_ids.device.index if device is None else device
    _native.check(lib.crag_merge_topk(int(dev or 0), d_ids.data_ptr(), d_scores.data_ptr(),
                                      d_counts.data_ptr(), n_lists, nq, k, d_out_ids.data_ptr(),
                                      d_out_scores.data_ptr(), d_out_counts.data_ptr(),
                                      ctypes.c_void_p(stream)), "crag_merge_topk")


class ResultRecord:
    """One rank's search output packed for a single all-gather: a uint8 CUDA buffer with typed views
    (ids int64 [nq,k], scores fp32 [nq,k], counts int32 [nq]) laid out as crag_merge_topk_packed
    expects."""

    @staticmethod
    def record_bytes(nq: int, k: int) -> int:
        return int(_native.load().crag_result_record_bytes(int(nq), int(k)))

    def __init__(self, nq: int, k: int, device, buf=None) -> None:
        """buf: an existing uint8 tensor of record_bytes(nq, k) bytes to lay the record over (this rank's slot of an
        all-gather buffer: the collective then runs in place); default: a buffer of its own."""
        self.nq, self.k = int(nq), int(k)
        self.nbytes = self.record_bytes(self.nq, self.k)
        if buf is None:
            buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        if buf.dtype != torch.uint8 or buf.numel() != self.nbytes or not buf.is_contiguous() or buf.data_ptr() % 8:
            raise ValueError("a ResultRecord needs a contiguous, 8-byte aligned uint8 buffer of record_bytes(nq, k) bytes")
        self.buf = buf
        a, b = self.nq * self.k * 8, self.nq * self.k * 12
        self.ids = self.buf[:a].view(torch.int64).view(self.nq, self.k)
        self.scores = self.buf[a:b].view(torch.float32).view(self.nq, self.k)
        self.counts = self.buf[b:b + self.nq * 4].view(torch.int32)


def merge_topk_packed(d_records, n_lists: int, nq: int, k: int, d_out_ids, d_out_scores, d_out_counts,
                      stream: int = 0) -> None:
    """Merge n_lists gathered ResultRecords (one contiguous uint8 CUDA tensor) into [nq, k]."""
    lib = _native.load()
    _native.check(lib.crag_merge_topk_packed(int(d_records.device.index or 0), d_records.data_ptr(), int(n_lists),
                                             int(nq), int(k), d_out_ids.data_ptr(), d_out_scores.data_ptr(),
                                             d_out_counts.data_ptr(), ctypes.c_void_p(stream)),
                  "crag_merge_topk_packed")
