"""Developer probe: crag_enc_wide_gemm (+ reduce) at 64 / 128 token rows against the library GEMM (torch.matmul ->
hipBLASLt) for the four projections of Qwen3-Embedding-4B: correctness and GPU time per call.  Timed as hipGraph
replays of 16 calls over 4 rotating weight copies (nothing cache resident; no host launch cost in the number: from Python
two launches cost 16 us of host time, more than either GEMM)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch.nn.functional as F
from cadence_rag_amd.encoder import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
shapes = [("qkv", 6144, 2560, False), ("o", 2560, 4096, False), ("gate_up", 19456, 2560, True), ("down", 2560, 9728, False)]
splits = {"qkv": (1, 2, 4), "o": (2, 4, 8), "gate_up": (1, 2), "down": (2, 4, 8, 19)}
COPIES, CALLS = 4, 16
def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(CALLS): fn(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(CALLS): fn(i)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (10 * CALLS) * 1e6
for m in (128, 64):
    tot_lib, tot_best = 0.0, 0.0
    for name, n, k, swi in shapes:
        ws = [(torch.randn(n, k, device=dev) * 0.02).bfloat16() for _ in range(COPIES)]
        wws = [ops.wide_gate_up_weight(w) if swi else ops.wide_weight(w) for w in ws]
        x = torch.randn(m, k, device=dev).bfloat16()
        out = torch.empty(m, n // 2 if swi else n, dtype=torch.bfloat16, device=dev)
        scratch = torch.empty(38 * 19456 * 128 // 4, dtype=torch.float32, device=dev)
        lin = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        ref = F.linear(x, ws[0]).float()
        if swi:
            g, u = ref[:, :n // 2].bfloat16().float(), ref[:, n // 2:].bfloat16().float()
            ref = (F.silu(g).bfloat16().float() * u)
        if swi:
            act = torch.empty(m, n // 2, dtype=torch.bfloat16, device=dev)
            lib_us = graph_time(lambda i: ops.swiglu(torch.matmul(x, ws[i % COPIES].t(), out=lin), act))
        else:
            lib_us = graph_time(lambda i: torch.matmul(x, ws[i % COPIES].t(), out=lin))
        line = f"m {m} {name:8s} [{n} x {k}] {n*k*2/1e6:6.1f} MB: library{' + swiglu' if swi else ''} {lib_us:6.1f} us ({n*k*2/lib_us/1e6:5.2f} TB/s)"
        best = 1e9
        for tile in ("128", "64"):
            os.environ["CRAG_WIDE_TILE"] = tile     # (developer switch: rows of W per workgroup; default: by workgroup count)
            for sk in splits[name]:
                ops.wide_gemm(x, wws[0], out, m, n, sk, swiglu=swi, scratch=scratch)
                err = float((out.float() - ref).abs().max()); scale = float(ref.abs().max())
                us = graph_time(lambda i: ops.wide_gemm(x, wws[i % COPIES], out, m, n, sk, swiglu=swi, scratch=scratch))
                best = min(best, us)
                line += f" | tile {tile} splitk {sk}: {us:5.1f} us err {err/scale:.0e}"
        os.environ.pop("CRAG_WIDE_TILE", None)
        tot_lib += lib_us; tot_best += best
        print(line, flush=True)
        del ws, wws
    print(f"m {m}: the four projections of a layer: library {tot_lib:.1f} us, wide_gemm (best split each) {tot_best:.1f} us", flush=True)
