"""Developer probe: the prefilter scan's exchange lags (CRAG_PF_LAGS = "derive,read": tiles between a publish of class
maxima and the delegates' derivation / every wave's read of the derived bounds): step, scan kernel, rest and the
candidates per query for k = 10 / 50 / 100 at 100 000 and 1M rows x 64 queries."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex
dev = torch.device("cuda", 0)
lags = sys.argv[1:] or ["2,4", "1,3", "2,3", "1,2"]
for rows in (100_000, 1_000_000):
    big = bench.synth(rows, 1234, dev)
    q = bench.synth(64, 4321, dev)
    for lag in lags:
        os.environ["CRAG_PF_LAGS"] = lag
        idx = DenseIndex(bench.DIM, capacity=rows, device=0); idx.add(big)
        for k in (10, 50, 100):
            leg = bench.search_leg(idx, q, k, 200, 20, 3, prewarm_s=0.1)
            st = leg["stats"]; n = max(st["searches"], 1)
            print(f"rows {rows} lags {lag} k {k}: step {min(leg['times'])/200*1e6:.1f} us, scan {leg['scan_us']:.1f}, "
                  f"rest {leg['rest_us']:.1f}, cand/q {st['candidates']/n/64:.0f}, rescored/q {st['rescored_rows']/n/64:.1f}", flush=True)
        idx.close()
    del big
