"""Developer: in-kernel timeline of the scan kernel (debug build with stamps)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CRAG_DEBUG_MODE"] = "64"
from cadence_rag_amd.dense_index import DenseIndex
from cadence_rag_amd import _native
rows = int(os.environ.get("ROWS", 100000)); nq = 32; k = 10
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
c = torch.randn(rows, 1024, generator=g); c /= c.norm(dim=1, keepdim=True); c = c.to(dev)
q = torch.randn(nq, 1024, generator=g).to(dev)
ix = DenseIndex(1024, rows); ix.add(c)
oi = torch.empty(nq, k, dtype=torch.int64, device=dev); os_ = torch.empty(nq, k, device=dev); oc = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10): ix.search_async(q, k, oi, os_, oc, stream=st)
torch.cuda.synchronize()
lib = _native.load()
lib.crag_debug_read_stamps.restype = ctypes.c_int
out = np.zeros((256, 4), dtype=np.uint64)
rc = lib.crag_debug_read_stamps(ix._h, ctypes.c_void_p(out.ctypes.data), 256)
t = out.astype(np.float64) / 100.0  # 100 MHz -> us
t0 = t[:, 0].min()
print("rc", rc)
print("WG start   us: min %.2f max %.2f" % (t[:,0].min()-t0, t[:,0].max()-t0))
print("prologue   us: avg %.2f max %.2f" % ((t[:,1]-t[:,0]).mean(), (t[:,1]-t[:,0]).max()))
print("loop       us: avg %.2f min %.2f max %.2f" % ((t[:,2]-t[:,1]).mean(), (t[:,2]-t[:,1]).min(), (t[:,2]-t[:,1]).max()))
print("drain+wr   us: avg %.2f max %.2f" % ((t[:,3]-t[:,2]).mean(), (t[:,3]-t[:,2]).max()))
print("WG end     us: min %.2f max %.2f" % (t[:,3].min()-t0, t[:,3].max()-t0))
