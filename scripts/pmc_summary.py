#!/usr/bin/env python3
"""Reduce rocprofv3 output to the small summaries kept under profiles/.

  pmc_summary.py pmc   <rocprof_out_dir> <out_csv>        per-kernel mean of every counter collected
  pmc_summary.py stats <rocprof_out_dir> <out_csv>        copy of the *_kernel_stats.csv
  pmc_summary.py traffic <fetch_csv> <write_csv> <ROWSxQUERIESxK> <traffic.json>
        fold the scan kernel's FETCH_SIZE / WRITE_SIZE means into profiles/traffic.json
        (FETCH_SIZE x2: gfx950 reports half of a wide coalesced streaming read,
         /opt/skills/guides/MI355X_MICROARCH.md HBM section; values are KB -> x1024).
"""
from __future__ import annotations

import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def _find(root: str, suffix: str) -> str:
    hits = sorted(glob.glob(os.path.join(root, "**", f"*{suffix}"), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {root}")
    return hits[-1]


def pmc(root: str, out_csv: str, skip: int = 10) -> None:
    src = _find(root, "counter_collection.csv")
    acc = defaultdict(list)
    with open(src, newline="") as fh:
        for row in csv.DictReader(fh):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "launches", "avg_value"])
        for (kern, ctr), vals in acc.items():
            steady = vals[skip:] if len(vals) > 2 * skip else vals
            w.writerow([kern, ctr, len(vals), f"{sum(steady) / len(steady):.2f}"])


def stats(root: str, out_csv: str) -> None:
    shutil.copyfile(_find(root, "kernel_stats.csv"), out_csv)


def _scan_mean(path: str, counter: str):
    """(kernel, mean) of `counter` for the dominant scan kernel of a run (the prefilter / scan kernel with the most
    launches)."""
    best = None
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["counter"] != counter:
                continue
            name = row["kernel"]
            if "prefilter_" in name or "scan_pipe" in name:
                key = (int(row["launches"]), "prefilter_" in name)
                if best is None or key > best[0]:
                    best = (key, name, float(row["avg_value"]))
    return (None, None) if best is None else (best[1], best[2])


def traffic(fetch_csv: str, write_csv: str, workload: str, out_json: str) -> None:
    """workload = ROWSxQUERIESxK as bench.py's roofline() keys it."""
    kern, fetch_kb = _scan_mean(fetch_csv, "FETCH_SIZE")
    _, write_kb = _scan_mean(write_csv, "WRITE_SIZE")
    doc = json.load(open(out_json)) if os.path.exists(out_json) else {}
    doc["method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                     "`python bench.py --steps .. --no-cpu-baseline --no-encode --no-target-1m [--rows-per-gpu 1000000 "
                     "--queries 32|64]` (scripts/gpu_round_check.sh); per-kernel means over launches 10.. "
                     "(scripts/pmc_summary.py); KB -> bytes x1024; FETCH_SIZE doubled (gfx950 reports half of a wide "
                     "coalesced streaming read, MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is.")
    rows, nq, k = (int(v) for v in workload.split("x"))
    doc.setdefault("by_workload", {})[workload] = {
        "kernel": kern.split("(")[0].replace("void ", "").strip(),
        "FETCH_SIZE_KB": fetch_kb,
        "WRITE_SIZE_KB": write_kb,
        "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + (write_kb or 0.0) * 1024),
        # bytes of a row the scan streams: 2 KiB from the fp16 mirror (prefilter_kernel<NQB, SETS, true, ..>), else the fp32 row
        "algorithmic_bytes_per_launch": rows * (2048 if re.search(r"prefilter_kernel<\d+, \d+, true", kern) else 4096)
                                        + nq * 4096 + nq * k * 12,
    }
    json.dump(doc, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    cmd = sys.argv[1]
    {"pmc": pmc, "stats": stats, "traffic": traffic}[cmd](*sys.argv[2:])
