"""Developer probe: does PyTorch's TunableOp (it times every hipBLASLt / rocBLAS algorithm for a GEMM shape once and
keeps the fastest) beat the libraries' default heuristic on the encoder's four GEMM shapes?"""
import time, torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
T = 65588
import os
M = int(os.environ.get("ROWS", 32768))
shapes = {"qkv": (M, 2560, 6144), "o": (M, 4096, 2560), "gate_up": (M, 2560, 19456), "down": (M, 9728, 2560)}
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
ops = {}
for name, (m, kin, kout) in shapes.items():
    x = torch.randn(m, kin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(kout, kin, device=dev, dtype=torch.bfloat16) * 0.02
    ops[name] = (x, w, 2.0 * m * kin * kout)
base = {}
for name, (x, w, fl) in ops.items():
    base[name] = bench(lambda: F.linear(x, w))
    print(f"default   {name:16s}: {fl / base[name] / 1e12:7.1f} TF  {base[name] * 1e3:.3f} ms", flush=True)
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(100); tn.set_max_tuning_iterations(10)
t0 = time.perf_counter()
for name, (x, w, fl) in ops.items():
    F.linear(x, w); torch.cuda.synchronize()
    print(f"tuned {name} after {time.perf_counter() - t0:.1f} s", flush=True)
tn.tuning_enable(False)
tot0 = tot1 = 0.0
for name, (x, w, fl) in ops.items():
    t1 = bench(lambda: F.linear(x, w))
    tot0 += base[name]; tot1 += t1
    print(f"tunableop {name:16s}: {fl / t1 / 1e12:7.1f} TF  {t1 * 1e3:.3f} ms  ({base[name] / t1:.3f}x)", flush=True)
print(f"per {M} rows and layer: {tot0 * 1e3:.3f} ms -> {tot1 * 1e3:.3f} ms")
tn.write_file(os.environ.get("OUT", "/tmp/tunableop_results.csv"))
try:
    print(tn.get_results()[:8])
except Exception as exc:
    print("results:", exc)
