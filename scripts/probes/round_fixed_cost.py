"""Developer probe: what does a timed round of K searches cost beyond K x (steady-state step)?  The driver times
`--steps 20`: 20 steps between two synchronisations.  Median over 30 rounds of: sync, t0, K x search_async, sync, t1."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex

dev = torch.device("cuda", 0)
rows, nq, k = 100_000, 64, 10
corpus = bench.synth(rows, 1234, dev)
idx = DenseIndex(bench.DIM, capacity=rows, device=0)
idx.add(corpus)
q = bench.synth(nq, 4321, dev)
oi = torch.empty(nq, k, dtype=torch.int64, device=dev); osc = torch.empty(nq, k, dtype=torch.float32, device=dev)
oc = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2000):
    idx.search_async(q, k, oi, osc, oc, stream=st)
torch.cuda.synchronize()
for sampling in (0, 8):
    idx.profile_enable(sampling)
    res = {}
    for K in (1, 2, 5, 10, 20, 40, 80, 320):
        ts, enq = [], []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                idx.search_async(q, k, oi, osc, oc, stream=st)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t2 - t0) * 1e6); enq.append((t1 - t0) * 1e6)
        res[K] = (statistics.median(ts), statistics.median(enq))
    base = (res[320][0] - res[80][0]) / 240
    print(f"event sampling every {sampling}: steady step {base:.2f} us")
    for K, (t, e) in res.items():
        print(f"  K={K:4d}: round {t:8.1f} us = {t / K:7.2f} per step; host enqueue of the K calls {e:7.1f} us; "
              f"round - K x steady = {t - K * base:6.1f} us", flush=True)
idx.profile_enable(0)
