"""HTTP shim with the reference gateway's contract, so an UNMODIFIED reference app can point
EMBEDDINGS_BASE_URL at the MI355X box: `POST /embed {texts, model?} -> {embeddings, model}` and
`GET /health` (/root/reference/P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:489-497,669-716).
The body is served by the in-process encoder registered with embeddings.set_encoder().
`POST /retrieve` is the reference's own route (/root/reference/app/main.py:184-186) over
retrieve.retrieve_evidence and the backend registered with retrieve.set_backend().

    uvicorn cadence_rag_amd.gateway:app --host 0.0.0.0 --port 8100
"""
from __future__ import annotations

from datetime import datetime
from typing import List, Literal, Optional
from uuid import UUID

from fastapi import FastAPI, HTTPException
from pydantic import BaseModel, Field

from . import embeddings
from . import retrieve as _retrieve
from .config import settings


class EmbedRequest(BaseModel):
    texts: List[str] = Field(default_factory=list)
    model: Optional[str] = None


class EmbedResponse(BaseModel):
    embeddings: List[List[float]]
    model: str


app = FastAPI(title="Cadence RAG MI355X embedding gateway")


@app.get("/health")
def health() -> dict:
    enc = embeddings.get_encoder()
    return {"status": "ok" if enc is not None else "degraded", "backend": "mi355x-native",
            "encoder_loaded": enc is not None, "embed_output_dim": settings.embeddings_dim,
            "model": settings.embeddings_model_id}


@app.post("/embed", response_model=EmbedResponse)
def embed(req: EmbedRequest) -> EmbedResponse:
    texts = [t for t in req.texts if isinstance(t, str) and t.strip()]
    if not texts:
        raise HTTPException(status_code=400, detail="texts must contain at least one non-empty string")
    enc = embeddings.get_encoder()
    if enc is None:
        raise HTTPException(status_code=502, detail="native encoder is not loaded")
    try:
        with embeddings._encoder_lock:
            vectors, model = enc.encode(texts)
    except Exception as exc:  # noqa: BLE001
        raise HTTPException(status_code=502, detail=f"encoder failed: {exc}") from exc
    vectors = [[float(x) for x in v] for v in vectors]
    if any(len(v) != settings.embeddings_dim for v in vectors):
        raise HTTPException(status_code=502, detail=f"encoder returned vectors of the wrong size "
                                                    f"(expected {settings.embeddings_dim})")
    return EmbedResponse(embeddings=vectors, model=req.model or model)


# ---- POST /retrieve: request models field-for-field app/schemas.py:71-93 -------------------------
class BudgetModel(BaseModel):
    max_evidence_items: int = 8
    max_total_chars: int = 6000


class RetrieveFiltersModel(BaseModel):
    date_from: Optional[datetime] = None
    date_to: Optional[datetime] = None
    call_ids: Optional[List[UUID]] = None
    external_id: Optional[str] = None
    external_source: Optional[str] = None
    call_tags: Optional[List[str]] = None


class RetrieveRequestModel(BaseModel):
    query: str
    intent: Literal["auto", "decision", "action_items", "who_said", "troubleshooting", "status"] = "auto"
    filters: Optional[RetrieveFiltersModel] = None
    budget: BudgetModel = Field(default_factory=BudgetModel)
    return_style: Literal["evidence_pack_json", "ids_only"] = "evidence_pack_json"
    debug: bool = False


@app.post("/retrieve")
def retrieve_endpoint(payload: RetrieveRequestModel) -> dict:
    filters = None
    if payload.filters is not None:
        filters = _retrieve.RetrieveFilters(**payload.filters.model_dump())
    request = _retrieve.RetrieveRequest(
        query=payload.query, intent=payload.intent, filters=filters,
        budget=_retrieve.Budget(**payload.budget.model_dump()), return_style=payload.return_style, debug=payload.debug)
    try:
        return _retrieve.retrieve_evidence(request)
    except RuntimeError as exc:  # no backend registered
        raise HTTPException(status_code=503, detail=str(exc)) from exc
