"""Developer probe: is a weight stream served faster from the Infinity Cache?  Back-to-back pairs, no idle time:
  A: sum(copy j) ; small_gemm(copy i)   with j != i  (the GEMM's weights come from HBM)
  B: sum(copy i) ; small_gemm(copy i)               (a default-policy read of the same bytes just before)
The streaming read is the same in both; (A - B) per pair is what the cache residency is worth to the GEMM."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16
forms = [("qkv", 2560, 6144, 12, False, True), ("o", 4096, 2560, 10, False, False),
         ("gate|up", 2560, 19456, 16, True, True), ("down", 9728, 2560, 10, False, False)]
m = 16
for name, k, n, rows, sw, pro in forms:
    ws = [(torch.randn(n, k, generator=g, device=dev) * 0.02).to(BF) for _ in range(6)]
    new_w = [ops.skinny_gate_up_weight(w) if sw else ops.small_weight(w, rows) for w in ws]
    del ws
    x = torch.randn(m, k, generator=g, device=dev).to(BF)
    d = torch.randn(m, k, generator=g, device=dev).to(BF)
    nw = torch.ones(k, device=dev, dtype=BF)
    res = torch.empty_like(x)
    out = torch.empty(m, n // 2 if sw else n, dtype=BF, device=dev)
    flat = [w.view(-1).view(torch.int16) for w in new_w]
    acc = torch.zeros((), dtype=torch.int64, device=dev)

    def gemm(i):
        if pro:
            ops.small_gemm(x, new_w[i % 6], out, m, n, rows, swiglu=sw, delta=d, norm_w=nw, res_out=res)
        else:
            ops.small_gemm(x, new_w[i % 6], out, m, n, rows)

    def loop(shift, only_sum=False, nrep=120):
        for i in range(12):
            torch.sum(flat[(i + shift) % 6], dim=(0,), out=acc); gemm(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(nrep):
            torch.sum(flat[(i + shift) % 6], dim=(0,), out=acc)
            if not only_sum:
                gemm(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nrep * 1e6

    s = loop(0, only_sum=True)
    a = loop(3)
    b = loop(0)
    print(f"{name:8s} {n * k * 2 / 1e6:6.1f} MB: sum alone {s:6.1f} us | sum(other)+gemm {a:6.1f} -> gemm {a - s:5.1f} us | "
          f"sum(same)+gemm {b:6.1f} -> gemm {b - s:5.1f} us", flush=True)
    del new_w, flat
