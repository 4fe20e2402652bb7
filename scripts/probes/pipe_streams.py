"""Developer probe: crag_index_search_pipelined with 1..4 internal streams (CRAG_PIPE_STREAMS) against the in-order form:
100 000 x 64, k = 10 and k = 100; 1M x 64, k = 10."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex
dev = torch.device("cuda", 0)
for rows in (100_000, 1_000_000):
    big = bench.synth(rows, 1234, dev); q = bench.synth(64, 4321, dev)
    for n in ("in-order", "1", "2", "3", "4"):
        if n != "in-order": os.environ["CRAG_PIPE_STREAMS"] = n
        idx = DenseIndex(bench.DIM, capacity=rows, device=0); idx.add(big)
        for k in ((10, 100) if rows == 100_000 else (10,)):
            leg = bench.search_leg(idx, q, k, 1000 if rows == 100_000 else 200, 20, 3, prewarm_s=0.1, pipelined=(n != "in-order"))
            steps = 1000 if rows == 100_000 else 200
            print(f"rows {rows} k {k} streams {n}: step {min(leg['times'])/steps*1e6:.1f} us, scan kernel {leg['scan_us']:.1f}", flush=True)
        idx.close()
    del big
