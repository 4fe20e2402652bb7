"""Developer probe: the configs[4] hybrid step alone, token lane in series vs on a side stream, in alternating rounds."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex
from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex

dev = torch.device("cuda", 0)
rows, nq = 1_000_000, 64
big = bench.synth(rows, 1234, dev)
idx = DenseIndex(bench.DIM, capacity=rows, device=0)
idx.add(big)
rng = np.random.default_rng(0)
vocab = np.array([f"TOK-{i}" for i in range(2000)])
n_tok = rng.integers(0, 4, size=rows)
flat = vocab[rng.integers(0, 2000, size=int(n_tok.sum()))].tolist()
row_tokens, o = [], 0
for n in n_tok.tolist():
    row_tokens.append(flat[o:o + n]); o += n
started = np.datetime64("2026-01-01", "us") + rng.integers(0, 365, size=rows).astype("timedelta64[D]")
tech = TechTokenIndex(row_tokens, np.arange(rows), started, dev, verify=False)
q = bench.synth(nq, 4321, dev)
qtoks = [vocab[rng.integers(0, 2000, size=3)].tolist() for _ in range(nq)]
bm25_ids = torch.from_numpy(rng.integers(0, rows, size=(nq, 50))).to(dev)
bm25_ct = torch.full((nq,), 50, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
hs = {name: HybridSearcher(idx, tech, dense_k=100, tech_k=50, overlap_lanes=ov) for name, ov in (("series", False), ("side", True))}


def timed(h, n):
    h.search(q, qtoks, (bm25_ids, bm25_ct), out_k=200, stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        h.search(q, qtoks, (bm25_ids, bm25_ct), out_k=200, stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for steps in (30, 100):
    for rnd in range(4):
        print(f"steps {steps} round {rnd}: " + ", ".join(f"{name} {timed(h, steps):.4f} ms" for name, h in hs.items()), flush=True)
