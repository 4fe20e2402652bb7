"""Embedding client seam — same module-level API as the reference's app/embeddings.py
(/root/reference/app/embeddings.py:11-100): EmbeddingClientError, EmbeddingResult,
embeddings_enabled, embed_texts, embed_texts_batched, with the same validation rules and error
strings (pinned by tests/golden/reference_host_logic.json, generated from the reference).

What changes is what stands behind embed_texts: instead of POSTing to an external Triton/ONNX
gateway, EMBEDDINGS_BASE_URL="native" routes to an in-process encoder registered with
set_encoder() (the MI355X Qwen3-Embedding encoder, cadence_rag_amd.encoder).  An http(s) URL
still works and speaks the reference's gateway protocol, so the two can be compared side by side.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import List, Optional, Protocol, Sequence, Tuple

from .config import settings


class EmbeddingClientError(RuntimeError):
    pass


@dataclass(frozen=True)
class EmbeddingResult:
    vectors: List[List[float]]
    model: str


@dataclass(frozen=True)
class DeviceEmbeddingResult:
    """embed_texts_device's result: `vectors` is a float32 [n, embeddings_dim] tensor that stays on the encoder's
    GPU (the backfill hands it to crag_index_add as a device pointer; nothing is converted to Python floats)."""
    vectors: "object"
    model: str


class Encoder(Protocol):
    """In-process backend: texts -> ([n][dim] vectors, model id).  May raise any exception; it is
    reported as EmbeddingClientError (the reference's callers only catch that type).  An encoder may also offer
    encode_device(texts) -> (float32 CUDA tensor [n, dim], model id)."""

    def encode(self, texts: Sequence[str]) -> Tuple[Sequence[Sequence[float]], str]: ...


_encoder: Optional[Encoder] = None
_encoder_lock = threading.Lock()  # FastAPI runs sync endpoints on a threadpool: one GPU submission at a time


def set_encoder(encoder: Optional[Encoder], warm: bool = True) -> None:
    """Register the in-process encoder.  warm: let it build what its short-query paths want NOW (Qwen3Encoder.warm_up:
    the re-tiled weight copies, seconds of work) instead of inside the first /retrieve request."""
    global _encoder
    _encoder = encoder
    if warm and encoder is not None and callable(getattr(encoder, "warm_up", None)):
        encoder.warm_up()


def get_encoder() -> Optional[Encoder]:
    return _encoder


def embeddings_enabled() -> bool:
    return bool(settings.embeddings_base_url.strip())


def _is_native(url: str) -> bool:
    return url.strip().lower().startswith("native")


def _validate_texts(texts: Sequence[str]) -> List[str]:
    kept = [t.strip() for t in texts if isinstance(t, str) and t.strip()]
    if not kept:
        raise EmbeddingClientError("embedding request requires at least one non-empty text")
    return kept


def _validate_vectors(vectors: Sequence[Sequence[float]]) -> List[List[float]]:
    want = settings.embeddings_dim
    out: List[List[float]] = []
    for i, vec in enumerate(vectors):
        if len(vec) != want:
            raise EmbeddingClientError(f"embedding {i} has dim {len(vec)}; expected {want}")
        out.append([float(x) for x in vec])
    return out


def _embed_native(cleaned: List[str]) -> Tuple[Sequence[Sequence[float]], str]:
    enc = _encoder
    if enc is None:
        raise EmbeddingClientError("native embedding encoder is not loaded (call set_encoder)")
    try:
        with _encoder_lock:
            return enc.encode(cleaned)
    except EmbeddingClientError:
        raise
    except Exception as exc:  # noqa: BLE001 - callers rely on a single error type
        raise EmbeddingClientError(f"native embedding encoder failed: {exc}") from exc


def _embed_http(cleaned: List[str]) -> Tuple[Sequence[Sequence[float]], str]:
    import httpx  # only needed for the gateway path

    url = settings.embeddings_base_url.rstrip("/") + "/embed"
    body = {"texts": cleaned, "model": settings.embeddings_model_id}
    try:
        with httpx.Client(timeout=httpx.Timeout(settings.embeddings_timeout_s)) as client:
            resp = client.post(url, json=body)
    except httpx.HTTPError as exc:
        raise EmbeddingClientError(f"embedding HTTP request failed: {exc}") from exc
    if resp.status_code != 200:
        detail = resp.text.strip()[:400]
        raise EmbeddingClientError(f"embedding service returned {resp.status_code}: {detail}")
    payload = resp.json()
    raw = payload.get("embeddings")
    if not isinstance(raw, list):
        raise EmbeddingClientError("embedding response missing 'embeddings' list")
    return raw, str(payload.get("model") or settings.embeddings_model_id)


def embed_texts(texts: Sequence[str]) -> EmbeddingResult:
    if not embeddings_enabled():
        raise EmbeddingClientError("EMBEDDINGS_BASE_URL is not configured")
    cleaned = _validate_texts(texts)
    if _is_native(settings.embeddings_base_url):
        raw, model = _embed_native(cleaned)
    else:
        raw, model = _embed_http(cleaned)
    if len(raw) != len(cleaned):
        raise EmbeddingClientError(
            f"embedding response count mismatch: got {len(raw)}, expected {len(cleaned)}")
    return EmbeddingResult(vectors=_validate_vectors(raw), model=model)


def embed_texts_device(texts: Sequence[str]) -> DeviceEmbeddingResult:
    """embed_texts for consumers that keep the vectors on the GPU (the HBM index sink of the backfill): same
    guards, text validation and error strings, but the native encoder's output tensor is returned as it is —
    no .tolist(), no per-float validation loop (the dim check is the tensor's shape)."""
    if not embeddings_enabled():
        raise EmbeddingClientError("EMBEDDINGS_BASE_URL is not configured")
    cleaned = _validate_texts(texts)
    enc = _encoder
    if not _is_native(settings.embeddings_base_url) or enc is None or not hasattr(enc, "encode_device"):
        raise EmbeddingClientError("device-resident embeddings need the native encoder (EMBEDDINGS_BASE_URL=native)")
    try:
        with _encoder_lock:
            vecs, model = enc.encode_device(cleaned)
    except EmbeddingClientError:
        raise
    except Exception as exc:  # noqa: BLE001 - callers rely on a single error type
        raise EmbeddingClientError(f"native embedding encoder failed: {exc}") from exc
    if int(vecs.shape[0]) != len(cleaned):
        raise EmbeddingClientError(
            f"embedding response count mismatch: got {int(vecs.shape[0])}, expected {len(cleaned)}")
    if vecs.dim() != 2 or int(vecs.shape[1]) != settings.embeddings_dim:
        raise EmbeddingClientError(f"embedding 0 has dim {int(vecs.shape[-1])}; expected {settings.embeddings_dim}")
    return DeviceEmbeddingResult(vectors=vecs, model=model)


def embed_texts_batched(texts: Sequence[str], batch_size: Optional[int] = None) -> EmbeddingResult:
    cleaned = _validate_texts(texts)
    size = batch_size or settings.embeddings_batch_size
    if size <= 0:
        raise EmbeddingClientError("batch size must be > 0")
    vectors: List[List[float]] = []
    model = settings.embeddings_model_id
    for lo in range(0, len(cleaned), size):
        part = embed_texts(cleaned[lo:lo + size])
        vectors.extend(part.vectors)
        model = part.model
    return EmbeddingResult(vectors=vectors, model=model)
