"""Settings for the dense lane — the EMBEDDINGS_* knobs of the reference
(/root/reference/app/config.py:10-16,25-26) with the same names, defaults and env-var spelling
(case-insensitive, no prefix).  Plain dataclass: the reference's pydantic-settings object is
mutated by its tests with monkeypatch.setattr(settings, ...), and so is this one.

New knob: EMBEDDINGS_BASE_URL keeps its meaning ("" disables the dense lane); the value
"native" (or "native://...") selects the in-process MI355X encoder instead of an HTTP gateway.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields


def _env(name: str, default):
    for key, raw in os.environ.items():
        if key.lower() == name.lower():
            if isinstance(default, bool):
                return raw.strip().lower() in ("1", "true", "yes", "on")
            return type(default)(raw)
    return default


@dataclass
class Settings:
    embeddings_base_url: str = ""
    embeddings_model_id: str = "Qwen/Qwen3-Embedding-4B"
    embeddings_dim: int = 1024
    embeddings_timeout_s: float = 180.0
    embeddings_batch_size: int = 32
    embeddings_exact_scan_threshold: int = 2000
    embeddings_hnsw_ef_search: int = 80
    ingest_auto_embed_on_success: bool = True
    ingest_auto_embed_fail_on_error: bool = False
    # native lane only
    embeddings_device: int = 0
    embeddings_max_length: int = 1024  # gateway truncation (RUNBOOK:484,748)

    @classmethod
    def from_env(cls) -> "Settings":
        return cls(**{f.name: _env(f.name, f.default) for f in fields(cls)})


settings = Settings.from_env()
