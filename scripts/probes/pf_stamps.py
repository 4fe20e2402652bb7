"""Developer probe: where a prefilter scan spends its time (s_memrealtime stamps per workgroup: kernel entry,
first tile reduced, tile loop done, kernel exit).   CRAG_PF_STAMPS=1 python scripts/probes/pf_stamps.py ROWS NQ"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["CRAG_PF_STAMPS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd import _native  # noqa: E402
from cadence_rag_amd.dense_index import DenseIndex  # noqa: E402

rows, nq = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
ix = DenseIndex(1024, capacity=rows)
for lo in range(0, rows, 131072):
    c = torch.randn(min(131072, rows - lo), 1024, generator=g, device=dev)
    ix.add(c / c.norm(dim=1, keepdim=True))
q = torch.randn(nq, 1024, generator=g, device=dev)
oi = torch.empty(nq, 10, dtype=torch.int64, device=dev)
osc = torch.empty(nq, 10, dtype=torch.float32, device=dev)
oc = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib = _native.load()
lib.crag_debug_pf_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.crag_debug_pf_stamps.restype = ctypes.c_int
G = 256
acc, tl = [], []
for rep in range(40):
    ix.search_async(q, 10, oi, osc, oc, stream=st)
    if rep >= 20:
        raw = np.zeros((1024, 4), dtype=np.uint64)
        assert lib.crag_debug_pf_stamps(ix._h, raw.ctypes.data, 1024) == 0
        buf = raw[:G]
        t = buf.astype(np.int64)
        tl.append((raw.reshape(-1)[1024:1024 + 128].astype(np.int64).reshape(8, 16) - t[:, 0].min()) / 100.0)
        t0 = t[:, 0].min()
        acc.append(np.stack([(t[:, 0] - t0), (t[:, 1] - t0), (t[:, 2] - t0), (t[:, 3] - t0)], axis=1) / 100.0)  # us
a = np.mean(np.stack(acc), axis=0)
names = ["entry", "first tile reduced", "tile loop done", "exit"]
for i, n in enumerate(names):
    print(f"{n:20s}: min {a[:, i].min():7.2f}  median {np.median(a[:, i]):7.2f}  max {a[:, i].max():7.2f} us after the first workgroup's entry")
m = np.mean(np.stack(tl), axis=0)
print("per-tile timeline (us after entry; reduced tile t) of workgroups 0, 32, ..., 224:")
for r in m:
    print("  " + " ".join(f"{v:6.1f}" for v in r))
ex = (raw.reshape(-1)[2560:2560 + 4 * G].astype(np.int64).reshape(G, 4) - t[:, 0].min()) / 100.0
for i, n in enumerate(["loads issued", "first loads landed (vmcnt 0)", "first barrier passed", "first MFMA phase done"]):
    print(f"{n:30s}: median {np.median(ex[:, i]):7.2f} us (last search)")
it1 = (raw.reshape(-1)[3584:3584 + 8 * 64].astype(np.int64).reshape(64, 8) - t[:, 0].min()) / 100.0
for i, n in enumerate(["iteration 1 starts", "its MFMA phase done", "barrier passed", "partials summed", "bounds sorted", "publishes + staging done"]):
    print(f"{n:30s}: median {np.median(it1[:, i]):7.2f} us (workgroups 0..63, last search)")
fl = raw.reshape(-1)[2048:2048 + G].astype(np.int64)
print("mid-scan flushes per workgroup (last search): count histogram", np.bincount((fl & 0xffff).astype(int))[:6],
      "| staged at the last flush (median)", int(np.median((fl >> 16) & 0xffffff)), "| tile of the last flush (median)", int(np.median(fl >> 40)))
tiles = (rows + 31) // 32
print(f"tiles per workgroup: {tiles / G:.2f}; per-tile time in the loop (median wg): "
      f"{np.median((a[:, 2] - a[:, 1])) / max(tiles / G - 1, 1):.2f} us")
