"""Developer probe: from a rocprofv3 --kernel-trace CSV, the average duration of each crag kernel of a search and the
idle gaps between consecutive kernels of the steady-state loop.   python scripts/probes/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
rows = [r for r in rows if r[2].startswith("crag::")]
tail = rows[len(rows) // 2:]          # steady state: the second half of the run
dur, gap_after = defaultdict(list), defaultdict(list)
for a, b in zip(tail, tail[1:]):
    dur[a[2]].append(a[1] - a[0])
    gap_after[a[2] + " -> " + b[2]].append(b[0] - a[1])
for k, v in dur.items():
    print(f"{k:60s} n={len(v):6d} avg {sum(v) / len(v) / 1e3:8.2f} us  min {min(v) / 1e3:8.2f}")
for k, v in gap_after.items():
    if len(v) > 20:
        print(f"gap {k:90s} n={len(v):6d} avg {sum(v) / len(v) / 1e3:7.2f} us")
