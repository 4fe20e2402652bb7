"""The bench contract is ONE JSON line the driver can parse; it keeps 8 KB of stdout (round 3's 24 KB line left the
driver's record with `parsed: null`).  bench.py prints the legs on a `DETAIL` line first and LAST a compact line:
this test builds that line from a full set of legs and holds it under 4 KB."""
import io
import json
import sys
from contextlib import redirect_stdout
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _full_line():
    # a complete line of an earlier round (every leg present, incl. its long prose fields): the worst case in size
    full = json.loads((ROOT / "profiles" / "r03_bench_line.json").read_text())
    full["config"]["per_rank_step_breakdown"] = [
        {"rank": r, "rows": 125000, "search_us": 61.2, "exchange_us": 22.4, "merge_us": 8.1, "scan_kernel_us": 43.0}
        for r in range(8)]
    full["config"]["speedup_vs_same_job_on_one_gpu"] = 3.8
    full["large_k"] = {"100000x64x100": {"ms_per_step": 0.07, "roofline": {"frac": 0.66}}}
    return full


def test_compact_line_is_small_and_complete():
    full = _full_line()
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT, len(text)
    back = json.loads(text)
    assert back == line
    for key in CONTRACT_KEYS:
        assert key in back, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_us"):
        assert key in back["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert "workload" in back["config"] and "model" not in back["config"]
    assert back["value"] == full["value"] and back["roofline"]["frac"] == full["roofline"]["frac"]
    assert back["summary"]["target_1m_q64_frac"] == full["target_1m"]["q64"]["roofline"]["frac"]
    assert back["summary"]["encode_chunks_per_s"] == full["encode"]["value"]


def test_compact_line_survives_missing_legs_and_long_strings():
    full = _full_line()
    for leg in ("target_1m", "hybrid", "query_path", "encode", "fp32_rows_scan", "cpu_baseline"):
        full.pop(leg)
    full["config"]["workload"] = "w" * 5000
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
    assert "cpu_baseline" not in line and line["roofline"]["bound"] == "hbm"


def test_emit_prints_the_compact_line_last(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(_full_line())
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("DETAIL {")
    last = json.loads(lines[-1])
    assert len(lines[-1]) < bench.COMPACT_LIMIT and "roofline" in last and "cpu_baseline" in last
    assert json.loads((tmp_path / "gpurun_out" / "bench_detail.json").read_text())["value"] == last["value"]
