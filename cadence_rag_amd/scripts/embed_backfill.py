"""`python -m cadence_rag_amd.scripts.embed_backfill` — same stdout lines and exit behaviour as
the reference CLI (/root/reference/app/scripts/embed_backfill.py:8-30)."""
from __future__ import annotations

from ..config import settings
from ..embedding_pipeline import run_embedding_backfill
from ..embeddings import EmbeddingClientError


def main() -> None:
    summary = run_embedding_backfill(batch_size=settings.embeddings_batch_size, source="embed_backfill")
    for table, n in summary.per_table.items():
        print(f"[embed_backfill] finished table={table} updated={n}")
    print("[embed_backfill] complete "
          f"rows_updated={summary.rows_updated} calls_touched={summary.calls_touched} "
          f"ingestion_runs_inserted={summary.ingestion_runs_inserted} model={summary.model_used}")


if __name__ == "__main__":
    try:
        main()
    except (EmbeddingClientError, RuntimeError) as exc:
        raise SystemExit(f"embed_backfill failed: {exc}") from exc
