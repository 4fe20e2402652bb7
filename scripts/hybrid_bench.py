"""BASELINE configs[4] rehearsal: hybrid /retrieve on the GPU — dense top-100 + exact-token lane top-50 +
externally supplied BM25 ranks (pg_search is not reproducible: SURVEY.md 8f) fused with RRF, batch = 64
queries over a 1M-chunk corpus.  Prints one JSON line (not the driver's bench contract)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cadence_rag_amd.dense_index import DenseIndex
from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex, rrf_fuse

rows = int(os.environ.get("ROWS", 1_000_000)); nq = 64
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
index = DenseIndex(1024, capacity=rows)
for lo in range(0, rows, 100_000):
    c = torch.randn(min(100_000, rows - lo), 1024, generator=g); c /= c.norm(dim=1, keepdim=True)
    index.add(c.to(dev), torch.arange(lo, lo + c.shape[0], dtype=torch.int64, device=dev))
q = torch.randn(nq, 1024, generator=g).to(dev)
rng = np.random.default_rng(0)
vocab = [f"TOK-{i}" for i in range(2000)]
row_tokens = [[vocab[t] for t in rng.integers(0, 2000, size=rng.integers(0, 4))] for _ in range(rows)]
started = np.datetime64("2026-01-01", "us") + rng.integers(0, 365, size=rows).astype("timedelta64[D]")
tech = TechTokenIndex(row_tokens, np.arange(rows), started, dev, verify=False)  # batch path: no host round trip
qtoks = [[vocab[t] for t in rng.integers(0, 2000, size=3)] for _ in range(nq)]
bm25_ids = torch.from_numpy(rng.integers(0, rows, size=(nq, 50))).to(dev)
bm25_ct = torch.full((nq,), 50, dtype=torch.int32, device=dev)
d_ids = torch.empty(nq, 100, dtype=torch.int64, device=dev); d_sc = torch.empty(nq, 100, device=dev); d_ct = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream

hybrid = HybridSearcher(index, tech, dense_k=100, tech_k=50)

def step():
    return hybrid.search(q, qtoks, (bm25_ids, bm25_ct), out_k=200, stream=st)

for _ in range(3): out = step()
torch.cuda.synchronize(); n = 20; t0 = time.perf_counter()
for _ in range(n): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
def lane_ms(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 3)
split = {"dense_top100_ms": lane_ms(lambda: index.search_async(q, 100, d_ids, d_sc, d_ct, stream=st)),
         "token_lane_top50_ms": lane_ms(lambda: tech.search(qtoks, 50, stream=st)),
         "rrf_fuse_ms": lane_ms(lambda: rrf_fuse([(bm25_ids, bm25_ct), (t_ids0, t_ct0), (d_ids, d_ct)], out_k=200, stream=st))
         if (globals().update(dict(zip(("t_ids0", "t_ct0"), tech.search(qtoks, 50, stream=st)))) is None) else None}
print(json.dumps({"workload": f"hybrid retrieve, {rows} chunks, batch {nq}: dense top-100 + token lane top-50 + bm25 ranks (given) -> RRF",
                  "ms_per_batch": round(dt * 1e3, 3), "queries_per_s": round(nq / dt, 1), "fused_counts_min": int(out['counts'].min()), "split": split}))
