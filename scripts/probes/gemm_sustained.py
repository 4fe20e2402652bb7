"""Developer probe: SUSTAINED rate (3 s loops, the part at its power limit) of the encoder's four projection GEMMs
through torch.matmul with hipBLASLt and with rocBLAS, at 65 536 and 32 768 rows.  Short-burst numbers
(gemm_backends.py, TunableOp) are taken at boost clocks and do not carry over to the forward (encode_clocks.py)."""
import os, sys, time
import torch

dev = torch.device("cuda", 0)
shapes = [("qkv", 2560, 6144), ("o", 4096, 2560), ("gate_up", 2560, 19456), ("down", 9728, 2560)]


def sustained(fn, flops, seconds=3.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    return flops * n / (time.perf_counter() - t0) / 1e12


for lib in ("hipblaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for m in (65536, 32768):
        tot_t = 0.0
        for name, k, n in shapes:
            a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            fl = 2.0 * m * k * n
            r_nt = sustained(lambda: torch.matmul(a, w.t(), out=out), fl)
            wt = w.t().contiguous()
            r_nn = sustained(lambda: torch.matmul(a, wt, out=out), fl)
            tot_t += fl / max(r_nt, r_nn) / 1e12
            print(f"{lib:9s} M={m} {name:8s} K={k} N={n}: x @ W^T {r_nt:7.1f} TFLOP/s | x @ Wt (pre-transposed) {r_nn:7.1f}", flush=True)
            del a, w, out, wt
        print(f"{lib:9s} M={m}: one layer's four GEMMs at the better layout: {tot_t * 1e3 * 65536 / m:.2f} ms per 65 536 rows", flush=True)
