import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import bench
from cadence_rag_amd.dense_index import DenseIndex
dev = torch.device("cuda", 0)
for rows in (100_000, 1_000_000):
    big = bench.synth(rows, 1234, dev); idx = DenseIndex(bench.DIM, capacity=rows, device=0); idx.add(big)
    q = bench.synth(64, 4321, dev)
    for k in (50, 100):
        leg = bench.search_leg(idx, q, k, 200, 20, 3, prewarm_s=0.1)
        print(f"rsplit {os.environ.get('CRAG_RSPLIT_DEV','default')} rows {rows} k {k}: step {min(leg['times'])/200*1e6:.1f} us, scan {leg['scan_us']:.1f}, rest {leg['rest_us']:.1f}", flush=True)
    idx.close(); del big
