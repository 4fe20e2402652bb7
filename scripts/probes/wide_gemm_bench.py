"""Developer probe: crag_enc_wide_gemm (+ reduce) at 64 / 128 token rows against the library GEMM (torch.matmul ->
hipBLASLt) for the four projections of Qwen3-Embedding-4B: correctness and time per call, weights rotated over several
copies so that nothing is cache resident."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch.nn.functional as F
from cadence_rag_amd.encoder import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
shapes = [("qkv", 6144, 2560, False), ("o", 2560, 4096, False), ("gate_up", 19456, 2560, True), ("down", 2560, 9728, False)]
splits = {"qkv": (1, 2, 4, 5), "o": (4, 8, 16), "gate_up": (1, 2), "down": (4, 8, 19)}
COPIES = 4
def timeit(fn, n=40):
    for _ in range(5): fn(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for m in (128, 64):
    for name, n, k, swi in shapes:
        ws = [(torch.randn(n, k, device=dev) * 0.02).bfloat16() for _ in range(COPIES)]
        wws = [ops.wide_gate_up_weight(w) if swi else ops.wide_weight(w) for w in ws]
        x = torch.randn(m, k, device=dev).bfloat16()
        out = torch.empty(m, n // 2 if swi else n, dtype=torch.bfloat16, device=dev)
        ref = F.linear(x, ws[0]).float()
        if swi:
            g, u = ref[:, :n // 2].bfloat16().float(), ref[:, n // 2:].bfloat16().float()
            ref = (F.silu(g).bfloat16().float() * u)
        lib_us = timeit(lambda i: F.linear(x, ws[i % COPIES]))
        line = f"m {m} {name:8s} [{n} x {k}] {n*k*2/1e6:6.1f} MB: library {lib_us:6.1f} us ({n*k*2/lib_us/1e6:5.2f} TB/s)"
        for sk in splits[name]:
            ops.wide_gemm(x, wws[0], out, m, n, sk, swiglu=swi)
            err = float((out.float() - ref).abs().max()); scale = float(ref.abs().max())
            us = timeit(lambda i: ops.wide_gemm(x, wws[i % COPIES], out, m, n, sk, swiglu=swi))
            line += f" | splitk {sk}: {us:6.1f} us ({n*k*2/us/1e6:5.2f} TB/s) err {err/scale:.1e}"
        print(line, flush=True)
        del ws, wws
