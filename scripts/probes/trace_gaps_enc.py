"""Developer probe: from a rocprofv3 --kernel-trace CSV of scripts/probes/small_encode_trace.py, the average duration of
every encoder kernel of the steady-state replays and the idle gap behind each.
   python scripts/probes/trace_gaps_enc.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        full = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        tmpl = re.search(r"<([0-9, ]+)>", full)
        name = re.split(r"[<(]", full)[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:40] + (tmpl.group(0) if tmpl else "")))
rows.sort()
tail = rows[len(rows) // 2:]
dur, gap = defaultdict(list), defaultdict(list)
for a, b in zip(tail, tail[1:]):
    dur[a[2]].append(a[1] - a[0])
    if b[0] - a[1] < 50000:       # not the host gap between two forwards
        gap[a[2]].append(b[0] - a[1])
tot_d = tot_g = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    gl = gap.get(k, [0])
    print(f"{k:70s} n={len(v):6d} avg {sum(v) / len(v) / 1e3:7.2f} us  min {min(v) / 1e3:7.2f} | gap behind it avg {sum(gl) / len(gl) / 1e3:6.2f} us")
    tot_d += sum(v)
    tot_g += sum(gl)
print(f"kernel time {tot_d / 1e6:.2f} ms, gaps {tot_g / 1e6:.2f} ms over the second half of the trace")
