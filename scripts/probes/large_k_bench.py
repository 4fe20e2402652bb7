"""Developer probe: top-50 / top-100 searches, 64 queries over 100 000 and 1M rows: step, scan kernel and the rest
(prep + selection launch).  [Used to compare 2 / 4 / 8 selection blocks per query (crag_api.hip: fin.rsplit) and the
selection's k-th search variants.]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex
dev = torch.device("cuda", 0)
for rows in (100_000, 1_000_000):
    big = bench.synth(rows, 1234, dev); idx = DenseIndex(bench.DIM, capacity=rows, device=0); idx.add(big)
    q = bench.synth(64, 4321, dev)
    for k in (50, 100):
        leg = bench.search_leg(idx, q, k, 200, 20, 3, prewarm_s=0.1)
        print(f"rows {rows} k {k}: step {min(leg['times'])/200*1e6:.1f} us, scan {leg['scan_us']:.1f}, rest {leg['rest_us']:.1f}", flush=True)
    idx.close(); del big
