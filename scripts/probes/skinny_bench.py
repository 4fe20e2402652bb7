"""Developer probe: crag_enc_skinny_gemm on the four projection shapes of the 4B encoder, 16 and 32 rows: us per launch
and weight-stream GB/s, next to torch's library GEMM on the same operands."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 2560, 6144, False), ("o", 4096, 2560, False), ("gate_up+swiglu", 2560, 19456, True), ("down", 9728, 2560, False)]
for m in (16, 32):
    tot_s = tot_t = 0.0
    for name, k, n, sw in shapes:
        # 8 different weight copies round-robin so that nothing is served from the Infinity Cache
        ws = [(torch.randn(n, k, generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
        wsw = [ops.skinny_gate_up_weight(w) if sw else ops.skinny_weight(w) for w in ws]
        x = torch.randn(m, k, generator=g, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
        def run_s(i): ops.skinny_gemm(x, wsw[i % 6], out, m, n, swiglu=sw)
        def run_t(i):
            y = torch.nn.functional.linear(x, ws[i % 6])
            if sw: ops.swiglu(y, out)
        res = {}
        for label, fn in (("skinny", run_s), ("torch", run_t)):
            for i in range(12): fn(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(240): fn(i)
            torch.cuda.synchronize(); res[label] = (time.perf_counter() - t0) / 240
        b = n * k * 2
        tot_s += res["skinny"]; tot_t += res["torch"]
        print(f"M={m:2d} {name:15s} K={k} N={n}: skinny {res['skinny']*1e6:7.1f} us = {b/res['skinny']/1e9:6.0f} GB/s | torch {res['torch']*1e6:7.1f} us = {b/res['torch']/1e9:6.0f} GB/s", flush=True)
        del ws, wsw
    print(f"M={m}: per layer skinny {tot_s*1e6:.1f} us, torch {tot_t*1e6:.1f} us (202 MB of weights: 25 us at 8 TB/s)", flush=True)
