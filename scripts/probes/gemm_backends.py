"""Developer probe: rocBLAS vs hipBLASLt (torch.backends.cuda.preferred_blas_library) for the encoder's GEMM
shapes at the bench's token count, with activations/weights of realistic scale."""
import time, torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
T = 65588
shapes = {"qkv(32768 rows)": (32768, 2560, 6144), "o": (T, 4096, 2560), "gate_up": (T, 2560, 19456), "down": (T, 9728, 2560)}
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, (m, kin, kout) in shapes.items():
        x = torch.randn(m, kin, device=dev, dtype=torch.bfloat16)
        w = torch.randn(kout, kin, device=dev, dtype=torch.bfloat16) * 0.02
        t1 = bench(lambda: F.linear(x, w))
        print(f"{lib:9s} {name:16s}: {2.0 * m * kin * kout / t1 / 1e12:7.1f} TF  {t1 * 1e3:.3f} ms", flush=True)
