"""Retrieval-quality metrics and gate for the dense lane — same definitions and CLI flags as the
reference's eval/run_eval.py:26-65 and eval/regression_gate.py:14-62 (values pinned by
tests/golden/reference_host_logic.json)."""
from .metrics import compute_metrics, dcg, gate_failures, load_jsonl  # noqa: F401
