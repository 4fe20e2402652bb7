#!/usr/bin/env python3
"""Reduce rocprofv3 output to the small summaries kept under profiles/.

  pmc_summary.py pmc   <rocprof_out_dir> <out_csv>        per-kernel mean of every counter collected
  pmc_summary.py stats <rocprof_out_dir> <out_csv>        copy of the *_kernel_stats.csv
  pmc_summary.py traffic <fetch_csv> <write_csv> <queries_per_step> <traffic.json> <alg_bytes>
        fold the scan kernel's FETCH_SIZE / WRITE_SIZE means into profiles/traffic.json
        (FETCH_SIZE x2: gfx950 reports half of a wide coalesced streaming read,
         /opt/skills/guides/MI355X_MICROARCH.md HBM section; values are KB -> x1024).
"""
from __future__ import annotations

import csv
import glob
import json
import os
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
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["counter"] == counter and "scan_" in row["kernel"]:
                val = row.get("avg_value") or row.get("avg_value_KB")
                return row["kernel"].split("(")[0], float(val)
    return None, None


def traffic(fetch_csv: str, write_csv: str, nq: str, out_json: str, alg_bytes: str) -> None:
    kern, fetch_kb = _scan_mean(fetch_csv, "FETCH_SIZE")
    _, write_kb = _scan_mean(write_csv, "WRITE_SIZE")
    doc = json.load(open(out_json)) if os.path.exists(out_json) else {}
    doc.setdefault("method", __doc__.strip().splitlines()[-3].strip())
    doc.setdefault("by_queries_per_step", {})[str(nq)] = {
        "kernel": kern,
        "FETCH_SIZE_KB": fetch_kb,
        "WRITE_SIZE_KB": write_kb,
        "scan_kernel_hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + (write_kb or 0.0) * 1024),
    }
    doc.setdefault("algorithmic_bytes_per_launch", {})[str(nq)] = int(alg_bytes)
    json.dump(doc, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    cmd = sys.argv[1]
    {"pmc": pmc, "stats": stats, "traffic": traffic}[cmd](*sys.argv[2:])
