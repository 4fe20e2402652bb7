"""Post-ingest auto-embed hook (B8).  The reference's ingest worker embeds a call right after its
ingest job succeeded (`_auto_embed_call_if_configured`, /root/reference/app/ingest_fs.py:809-837,
called at :894) and reports one of three status dictionaries; this is the native-lane counterpart,
written against that contract (the five cases of the reference's tests/unit/test_ingest_fs.py:130-195):

    hook off                      -> {"status": "skipped", "reason": "disabled"}
    no encoder configured         -> {"status": "skipped", "reason": "embeddings_not_configured"}
    backfill ran                  -> {"status": "ok", <counters of the BackfillSummary>}
    backfill raised, fail-open    -> {"status": "error", "error": str(exc)}
    backfill raised, fail-closed  -> the exception propagates (INGEST_AUTO_EMBED_FAIL_ON_ERROR)
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional
from uuid import UUID

from .config import settings
from .embedding_pipeline import run_embedding_backfill
from .embeddings import EmbeddingClientError, embeddings_enabled

logger = logging.getLogger(__name__)

_SOURCE = "ingest_auto_embed"
_OK_FIELDS = ("rows_updated", "calls_touched", "model_used", "ingestion_runs_inserted")


def _skip_reason() -> Optional[str]:
    if not settings.ingest_auto_embed_on_success:
        return "disabled"
    if not embeddings_enabled():
        return "embeddings_not_configured"
    return None


def _auto_embed_call_if_configured(call_id: UUID) -> Dict[str, Any]:
    reason = _skip_reason()
    if reason is not None:
        return {"status": "skipped", "reason": reason}
    batch = max(1, int(settings.embeddings_batch_size))
    try:
        summary = run_embedding_backfill(batch_size=batch, call_id=call_id, source=_SOURCE)
    except Exception as exc:  # noqa: BLE001 - an embedding failure must not take the ingest worker down
        if settings.ingest_auto_embed_fail_on_error:
            raise
        if not isinstance(exc, EmbeddingClientError):  # client errors are expected; anything else gets a trace
            logger.exception("ingest_job.auto_embed_failed call_id=%s error=%s", call_id, exc)
        return {"status": "error", "error": str(exc)}
    return {"status": "ok", **{name: getattr(summary, name) for name in _OK_FIELDS}}
