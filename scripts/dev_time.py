"""Developer timing of the search path (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cadence_rag_amd.dense_index import DenseIndex
rows = int(os.environ.get("ROWS", 100000)); nq = int(os.environ.get("NQ", 32)); k = int(os.environ.get("K", 10))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
c = torch.randn(rows, 1024, generator=g); c /= c.norm(dim=1, keepdim=True); c = c.to(dev)
q = torch.randn(nq, 1024, generator=g).to(dev)
ix = DenseIndex(1024, rows); ix.add(c)
oi = torch.empty(nq, k, dtype=torch.int64, device=dev); os_ = torch.empty(nq, k, device=dev); oc = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(20): ix.search_async(q, k, oi, os_, oc, stream=st)
torch.cuda.synchronize()
ix.profile_enable(True)
t0 = time.perf_counter()
for _ in range(200): ix.search_async(q, k, oi, os_, oc, stream=st)
torch.cuda.synchronize()
el = time.perf_counter() - t0
n, scan, merge = ix.profile_read()
print(f"mode={os.environ.get('CRAG_UNPIPELINED','0')} rows={rows} nq={nq} k={k}: step={el/200*1e6:.1f}us scan={scan/n*1e3:.1f}us merge={merge/n*1e3:.1f}us  scanBW={rows*4096/(scan/n*1e-3)/1e12:.2f}TB/s")
