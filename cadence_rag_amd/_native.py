"""ctypes binding of libcrag_dense.so (C ABI: include/crag_dense.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, every
entry point raises NativeLibraryError loudly.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path
from typing import Optional

_CSRC = Path(__file__).resolve().parent / "csrc"
# CRAG_DENSE_LIB: deployments that install the library elsewhere
LIB_PATH = Path(os.environ["CRAG_DENSE_LIB"]) if os.environ.get("CRAG_DENSE_LIB") else _CSRC / "libcrag_dense.so"

CRAG_MAX_K = 128
CRAG_DIM = 1024

# every symbol include/crag_dense.h declares: name -> (restype, argtypes)
_c = ctypes
_P = _c.c_void_p
SIGNATURES = {
    "crag_last_error": (_c.c_char_p, []),
    "crag_version": (_c.c_char_p, []),
    "crag_device_count": (_c.c_int, []),
    "crag_index_create": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int64, _c.POINTER(_P)]),
    "crag_index_destroy": (_c.c_int, [_P]),
    "crag_index_add": (_c.c_int, [_P, _P, _P, _c.c_int64]),
    "crag_index_update": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64]),
    "crag_index_size": (_c.c_int64, [_P]),
    "crag_index_capacity": (_c.c_int64, [_P]),
    "crag_index_dim": (_c.c_int, [_P]),
    "crag_index_get_rows": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P]),
    "crag_index_count_eligible": (_c.c_int, [_P, _P, _c.POINTER(_c.c_int64)]),
    "crag_index_search": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P, _P]),
    "crag_index_search_async": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P, _P, _P]),
    "crag_merge_topk": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    "crag_result_record_bytes": (_c.c_int64, [_c.c_int, _c.c_int]),
    "crag_merge_topk_packed": (_c.c_int, [_c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    "crag_rrf_fuse": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P]),
    "crag_tech_lane": (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P, _P, _P]),
    "crag_upload_slot_create": (_P, []),
    "crag_upload_slot_destroy": (None, [_P]),
    "crag_tech_lane_host": (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P, _P,
                                       _P, _P]),
    "crag_index_profile_enable": (_c.c_int, [_P, _c.c_int]),
    "crag_index_profile_read": (_c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_double),
                                           _c.POINTER(_c.c_double)]),
    "crag_index_profile_read_ex": (_c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_double),
                                              _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "crag_index_last_scan_kernel": (_c.c_char_p, [_P]),
    "crag_index_prefilter_row_bytes": (_c.c_int64, [_P]),
    "crag_index_search_pipelined": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P, _P, _P, _c.c_int]),
    "crag_index_join": (_c.c_int, [_P, _P]),
    "crag_index_phase_trace": (_c.c_int, [_P, _c.POINTER(_c.c_uint64)]),
    "crag_index_prefilter_stats": (_c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64),
                                              _c.POINTER(_c.c_int64)]),
    "crag_index_scan_geometry": (_c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int),
                                            _c.POINTER(_c.c_int), _c.POINTER(_c.c_int64)]),
    # include/crag_encoder.h
    "crag_enc_embed_gather": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int64, _P]),
    "crag_enc_rmsnorm": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_float, _P]),
    "crag_enc_qk_norm_rope": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_float, _P]),
    "crag_enc_v_transpose": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _P]),
    "crag_enc_attention": (_c.c_int, [_P, _P, _P, _P, _P, _P, _P, _c.c_int, _c.c_int64, _c.c_int, _c.c_int,
                                      _c.c_float, _P]),
    "crag_enc_swiglu": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _P]),
    "crag_enc_skinny_gemm": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "crag_enc_qk_rope_vt": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_float, _P, _P, _c.c_int64, _P]),
    "crag_enc_pool_normalize": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "crag_enc_pool_normalize_add": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "crag_enc_pool_normalize_rows": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "crag_enc_small_gemm": (_c.c_int, [_P, _P, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_float, _P]),
    "crag_enc_wide_partial_bytes": (_c.c_int64, [_c.c_int, _c.c_int, _c.c_int]),
    "crag_enc_wide_gemm": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "crag_enc_wide_gemm_direct": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "crag_enc_wide_reduce": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "crag_enc_wide_gemm_rows": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "crag_enc_rmsnorm_partials": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_float, _P]),
    "crag_enc_small_attention": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_float,
                                            _c.c_float, _P]),
    "crag_enc_small_attention_seqs": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int,
                                                 _c.c_int, _c.c_float, _c.c_float, _P]),
    "crag_enc_small_attention_seqs_parts": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _P, _P, _c.c_int, _P, _P, _c.c_int,
                                                       _c.c_int, _P, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _P]),
}


class NativeLibraryError(RuntimeError):
    """libcrag_dense.so is missing, failed to load, or a call into it failed."""


_lib: Optional[ctypes.CDLL] = None


def build_native(force: bool = False) -> Path:
    """Compile libcrag_dense.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_CSRC)] + (["-B"] if force else [])
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise NativeLibraryError(f"building libcrag_dense.so failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the dense lane has no CPU fallback. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {_CSRC}`.")
    try:
        lib = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:
        raise NativeLibraryError(f"failed to load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().crag_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeLibraryError(f"{what} failed (code {rc}): {last_error()}")
