"""ctypes wrapper around oracle/exact_scan.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (see the header of exact_scan.c for the parity status: "parity
unpinned", restating /root/reference/app/retrieve.py:339-353 + pgvector 0.8.1).
The product package (cadence_rag_amd/) never imports it.
"""
from __future__ import annotations

import ctypes
import hashlib
import platform
import subprocess
from pathlib import Path
from typing import Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
F32SEQ = 0  # pgvector-faithful fp32 sequential accumulation
F64 = 1  # fp64 accumulation ("truth")

_libs: dict = {}


def _host_tag() -> str:
    try:
        cpu = Path("/proc/cpuinfo").read_text().split("model name", 2)[1].split("\n", 1)[0]
    except Exception:  # pragma: no cover
        cpu = platform.processor()
    return hashlib.sha1((platform.machine() + cpu).encode()).hexdigest()[:10]


def build(fast: bool = False) -> Path:
    """Compile the oracle with gcc (strict: portable flags; fast: -march=native, per host)."""
    src = _HERE / "exact_scan.c"
    if not fast:
        out = _HERE / "liboracle_scan.so"
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off",
               "-o", str(out), str(src), "-lm"]
    else:
        outdir = _HERE / "_build" / _host_tag()
        outdir.mkdir(parents=True, exist_ok=True)
        out = outdir / "liboracle_scan_fast.so"
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-march=native",
               "-ftree-vectorize", "-fassociative-math", "-fno-signed-zeros",
               "-fno-trapping-math", "-fopenmp", "-o", str(out), str(src), "-lm"]
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(cmd, check=True)
    return out


def _lib(fast: bool = False) -> ctypes.CDLL:
    key = "fast" if fast else "strict"
    if key not in _libs:
        lib = ctypes.CDLL(str(build(fast)))
        lib.crag_oracle_topk.restype = ctypes.c_int
        lib.crag_oracle_topk.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ]
        lib.crag_oracle_scores.restype = ctypes.c_int
        lib.crag_oracle_scores.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_void_p,
        ]
        lib.crag_oracle_cosine_distance.restype = ctypes.c_double
        lib.crag_oracle_cosine_distance.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.crag_oracle_num_threads.restype = ctypes.c_int
        _libs[key] = lib
    return _libs[key]


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def exact_topk(queries, corpus, k: int, *, ids=None, mask=None, mode: int = F64,
               fast: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Exact cosine top-k. Returns (ids[nq,k] int64 (-1 pad), scores[nq,k] f64 (NaN pad), counts[nq]).

    mask: None, uint8 [ceil(n/8)] (shared) or [nq, ceil(n/8)] (per query); bit i&7 of byte i>>3.
    """
    q = _f32(queries)
    c = _f32(corpus)
    if q.ndim == 1:
        q = q[None, :]
    nq, dim = q.shape
    n = c.shape[0] if c.size else 0
    if n:
        assert c.shape[1] == dim
    ids_arr = None if ids is None else np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
    stride = 0
    mask_arr = None
    if mask is not None:
        mask_arr = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        if mask_arr.ndim == 2:
            assert mask_arr.shape[0] == nq
            stride = mask_arr.shape[1]
    out_ids = np.empty((nq, k), dtype=np.int64)
    out_scores = np.empty((nq, k), dtype=np.float64)
    out_counts = np.empty((nq,), dtype=np.int32)
    rc = _lib(fast).crag_oracle_topk(
        mode, q.ctypes.data, nq, c.ctypes.data if n else None, n, dim,
        None if ids_arr is None else ids_arr.ctypes.data,
        None if mask_arr is None else mask_arr.ctypes.data, stride, k,
        out_ids.ctypes.data, out_scores.ctypes.data, out_counts.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"crag_oracle_topk failed rc={rc}")
    return out_ids, out_scores, out_counts


def scores(queries, corpus, mode: int = F64) -> np.ndarray:
    q = _f32(queries)
    c = _f32(corpus)
    out = np.empty((q.shape[0], c.shape[0]), dtype=np.float64)
    rc = _lib().crag_oracle_scores(mode, q.ctypes.data, q.shape[0], c.ctypes.data,
                                   c.shape[0], q.shape[1], out.ctypes.data)
    if rc != 0:
        raise RuntimeError("crag_oracle_scores failed")
    return out


def cosine_distance(a, b, mode: int = F32SEQ) -> float:
    a = _f32(a)
    b = _f32(b)
    return float(_lib().crag_oracle_cosine_distance(mode, a.shape[0], a.ctypes.data, b.ctypes.data))


def num_threads(fast: bool = True) -> int:
    return int(_lib(fast).crag_oracle_num_threads())


def set_threads(n: int, fast: bool = True) -> None:
    _lib(fast).crag_oracle_set_threads(int(n))
