"""Developer probe: does crag_enc_small_gemm run faster when its weights were just read by another kernel (i.e. are
they served from the 256 MB Infinity Cache)?  The question behind it: a side branch of the encoder's graph could read
the NEXT projection's weights while the current projection runs, so that HBM never idles across kernel boundaries.
Per form, HIP-event time of one launch (median of 60; the event pair's own cost is printed and is the same for all):
  cold        six weight copies round robin (600 MB - 1.2 GB between two uses of a copy)
  warm        the same copy every launch
  prefetched  six copies round robin, each read by a plain streaming kernel (torch.sum) right before the launch
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16


def ev_time(fn, pre=None, n=60):
    ts = []
    for i in range(n + 8):
        if pre is not None:
            pre(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        if i >= 8:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


empty = ev_time(lambda i: None)
print(f"event pair with nothing between: {empty:.1f} us", flush=True)
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

    def run(i, one=False):
        w = new_w[0 if one else i % 6]
        if pro:
            ops.small_gemm(x, w, out, m, n, rows, swiglu=sw, delta=d, norm_w=nw, res_out=res)
        else:
            ops.small_gemm(x, w, out, m, n, rows)

    cold = ev_time(run)
    warm = ev_time(lambda i: run(i, True))
    pref = ev_time(run, pre=lambda i: flat[i % 6].sum())
    touch = ev_time(lambda i: flat[i % 6].sum())
    mb = n * k * 2 / 1e6
    print(f"{name:8s} {mb:6.1f} MB: cold {cold:6.1f} us | warm {warm:6.1f} | prefetched {pref:6.1f} | "
          f"(the streaming read alone {touch:6.1f} us); minus the event pair: cold {cold - empty:5.1f} "
          f"({mb / (cold - empty) / 1e3:.2f} TB/s), warm {warm - empty:5.1f}, prefetched {pref - empty:5.1f}", flush=True)
    del new_w, flat
