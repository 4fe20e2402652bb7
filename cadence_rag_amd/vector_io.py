"""pgvector interchange for the corpus: the text form `[v0,v1,...]` the reference builds with
_vector_literal (/root/reference/app/retrieve.py:263-264, embedding_pipeline.py:63-64: '.10g' per
component, parsed by pgvector into float4) and the binary COPY/send form of `vector`
(int16 dim, int16 unused, dim x float4, network byte order — pgvector 0.8.1 vector_send/recv).
Used to load `SELECT id, embedding FROM chunks WHERE embedding IS NOT NULL` into a DenseIndex and to
write embeddings back in bulk instead of one 15 KB literal per UPDATE."""
from __future__ import annotations

import struct
from typing import Iterable, List, Sequence

import numpy as np


def format_vector(values: Sequence[float]) -> str:
    return "[" + ",".join(format(float(v), ".10g") for v in values) + "]"


def parse_vector(text: str, dim: int | None = None) -> np.ndarray:
    body = text.strip()
    if len(body) < 2 or body[0] != "[" or body[-1] != "]":
        raise ValueError("vector literal must look like '[v0,v1,...]'")
    inner = body[1:-1].strip()
    vals = np.array([float(x) for x in inner.split(",")] if inner else [], dtype=np.float32)
    if dim is not None and vals.size != dim:
        raise ValueError(f"expected {dim} dimensions, not {vals.size}")
    return vals


def parse_vectors(texts: Iterable[str], dim: int) -> np.ndarray:
    rows = [parse_vector(t, dim) for t in texts]
    return np.stack(rows) if rows else np.zeros((0, dim), dtype=np.float32)


def to_binary(values: Sequence[float]) -> bytes:
    v = np.asarray(values, dtype=">f4")
    if v.ndim != 1 or v.size > 16000:
        raise ValueError("vector must be 1-D with at most 16000 dimensions")
    return struct.pack(">hh", v.size, 0) + v.tobytes()


def from_binary(buf: bytes) -> np.ndarray:
    dim, unused = struct.unpack(">hh", buf[:4])
    if unused != 0 or dim < 1 or len(buf) != 4 + 4 * dim:
        raise ValueError("malformed pgvector binary value")
    return np.frombuffer(buf, dtype=">f4", offset=4, count=dim).astype(np.float32)


def copy_rows_text(ids: Sequence[int], vectors: np.ndarray) -> List[str]:
    """Rows for `COPY tmp(id, embedding) FROM STDIN` (text format): 'id<TAB>[v,...]'."""
    return [f"{int(i)}\t{format_vector(v)}" for i, v in zip(ids, vectors)]
