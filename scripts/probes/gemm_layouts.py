"""Developer probe: hipBLASLt bf16 GEMM rate for the encoder's four linear shapes, weight stored [out,in]
(F.linear) vs [in,out] (matmul), and for a few token-chunk sizes."""
import time, torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
T = 65536
shapes = {"qkv": (2560, 6144), "o": (4096, 2560), "gate_up": (2560, 19456), "down": (9728, 2560)}
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for name, (kin, kout) in shapes.items():
    x = torch.randn(T, kin, device=dev, dtype=torch.bfloat16)
    w_oi = torch.randn(kout, kin, device=dev, dtype=torch.bfloat16) * 0.02
    w_io = w_oi.t().contiguous()
    out = torch.empty(T, kout, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * kin * kout
    t1 = bench(lambda: F.linear(x, w_oi))
    t2 = bench(lambda: torch.matmul(x, w_io))
    t3 = bench(lambda: torch.mm(x, w_io, out=out))
    line = f"{name:8s} K={kin:5d} N={kout:5d}: linear[out,in] {fl/t1/1e12:7.1f} TF  matmul[in,out] {fl/t2/1e12:7.1f} TF  mm(out=) {fl/t3/1e12:7.1f} TF"
    for chunk in (8192, 16384, 32768):
        def run():
            for lo in range(0, T, chunk): F.linear(x[lo:lo + chunk], w_oi)
        line += f"  chunk{chunk}: {fl/bench(run)/1e12:7.1f}"
    print(line, flush=True)
