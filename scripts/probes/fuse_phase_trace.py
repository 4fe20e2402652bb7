"""Developer probe: where rrf_fuse_kernel spends its time for the hybrid step's shape (64 queries; lanes of 50 BM25,
50 token-lane and 100 dense ids with overlaps; out_k 200).  Needs the trace build of the library:
   cd cadence_rag_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCRAG_FUSE_TRACE -c crag_fusion.hip \
      -o /tmp/f.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/probes/_trace/libcrag_dense_trace.so \
      crag_search.o crag_api.o crag_encoder.o crag_encoder_small.o crag_encoder_wide.o /tmp/f.o
   CRAG_DENSE_LIB=scripts/probes/_trace/libcrag_dense_trace.so python scripts/probes/fuse_phase_trace.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd import _native
from cadence_rag_amd.fusion import rrf_fuse

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
nq = 64
pool = torch.stack([torch.randperm(400, generator=g, device=dev) for _ in range(nq)]).to(torch.int64) + 1000
lanes = []
for width, lo in ((50, 0), (50, 30), (100, 60)):     # overlapping windows of each query's pool: shared keys
    ids = pool[:, lo:lo + width].contiguous()
    lanes.append((ids, torch.full((nq,), width, dtype=torch.int32, device=dev)))
out = None
for _ in range(20):
    out = rrf_fuse(lanes, out_k=200, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    rrf_fuse(lanes, out_k=200, out=out)
e1.record()
torch.cuda.synchronize()
print(f"rrf_fuse, 64 queries x (50 + 50 + 100) ids: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (stream time, host-paced)")
lib = _native.load()
if hasattr(lib, "crag_fuse_trace_read"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.crag_fuse_trace_read.restype = ctypes.c_int
    assert lib.crag_fuse_trace_read(buf) == 0
    t = list(buf)
    names = ["counts + table init", "insert (ids loaded, keys claimed)", "scores summed lane by lane", "ranks + stores", "padding"]
    print("  " + ", ".join(f"{nm} {(t[i + 1] - t[i]) / 100:.2f} us" for i, nm in enumerate(names)) + f" | total {(t[5] - t[0]) / 100:.2f} us")
