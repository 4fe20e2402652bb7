"""Post-ingest auto-embed hook — counterpart of `_auto_embed_call_if_configured`
(/root/reference/app/ingest_fs.py:809-837, called at :894 after a successful ingest job): embed the
just-ingested call with the native lane, fail-open by default, fail-closed when
INGEST_AUTO_EMBED_FAIL_ON_ERROR is set.  Same status dictionaries as the reference."""
from __future__ import annotations

import logging
from typing import Any, Dict
from uuid import UUID

from .config import settings
from .embedding_pipeline import run_embedding_backfill
from .embeddings import EmbeddingClientError, embeddings_enabled

logger = logging.getLogger(__name__)


def _auto_embed_call_if_configured(call_id: UUID) -> Dict[str, Any]:
    if not settings.ingest_auto_embed_on_success:
        return {"status": "skipped", "reason": "disabled"}
    if not embeddings_enabled():
        return {"status": "skipped", "reason": "embeddings_not_configured"}
    try:
        summary = run_embedding_backfill(batch_size=max(1, int(settings.embeddings_batch_size)),
                                         call_id=call_id, source="ingest_auto_embed")
    except EmbeddingClientError as exc:
        if settings.ingest_auto_embed_fail_on_error:
            raise
        return {"status": "error", "error": str(exc)}
    except Exception as exc:  # noqa: BLE001 - the worker must survive an embedding failure
        if settings.ingest_auto_embed_fail_on_error:
            raise
        logger.exception("ingest_job.auto_embed_failed call_id=%s error=%s", call_id, exc)
        return {"status": "error", "error": str(exc)}
    return {"status": "ok", "rows_updated": summary.rows_updated, "calls_touched": summary.calls_touched,
            "model_used": summary.model_used, "ingestion_runs_inserted": summary.ingestion_runs_inserted}
