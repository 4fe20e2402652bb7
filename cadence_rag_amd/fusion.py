"""GPU reciprocal-rank fusion of retrieval lanes (hybrid /retrieve, BASELINE configs[4]) — the
device counterpart of _rrf_merge (/root/reference/app/retrieve.py:245-260, host mirror:
cadence_rag_amd.retrieve._rrf_merge)."""
from __future__ import annotations

import ctypes
from typing import Dict, Sequence, Tuple

import torch

from . import _native

DEFAULT_RRF_K = 60


def rrf_fuse(lanes: Sequence[Tuple[torch.Tensor, torch.Tensor]], out_k: int, rrf_k: int = DEFAULT_RRF_K,
             stream: int = 0) -> Dict[str, torch.Tensor]:
    """lanes: [(ids int64 [nq, width] CUDA, counts int32 [nq] CUDA), ...] in lane order (the
    reference's order is bm25, tech_tokens, dense).  Returns ids [nq, out_k] (-1 pad), scores fp64,
    lane-hit masks (bit l = lane l), counts."""
    lib = _native.load()
    n = len(lanes)
    nq = int(lanes[0][0].shape[0])
    dev = lanes[0][0].device
    ids_arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(t.contiguous().data_ptr()) for t, _ in lanes])
    cnt_arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(c.contiguous().data_ptr()) for _, c in lanes])
    width = (ctypes.c_int * n)(*[int(t.shape[1]) for t, _ in lanes])
    out = {
        "ids": torch.empty(nq, out_k, dtype=torch.int64, device=dev),
        "scores": torch.empty(nq, out_k, dtype=torch.float64, device=dev),
        "lanes": torch.empty(nq, out_k, dtype=torch.int32, device=dev),
        "counts": torch.empty(nq, dtype=torch.int32, device=dev),
    }
    _native.check(lib.crag_rrf_fuse(n, ids_arr, cnt_arr, width, nq, int(rrf_k), int(out_k), out["ids"].data_ptr(),
                                    out["scores"].data_ptr(), out["lanes"].data_ptr(), out["counts"].data_ptr(),
                                    ctypes.c_void_p(stream)), "crag_rrf_fuse")
    return out
