"""Developer probe: the configs[4] hybrid step (bench.hybrid_leg) alone: token lane on a side stream vs lanes in series."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex

dev = torch.device("cuda", 0)
rows = 1_000_000
big = bench.synth(rows, 1234, dev)
idx = DenseIndex(bench.DIM, capacity=rows, device=0)
idx.add(big)
for _ in range(2):
    out = bench.hybrid_leg(idx, rows, dev, steps=60)
    print(json.dumps({k: v for k, v in out.items() if k not in ("dense_roofline", "workload")}), flush=True)
