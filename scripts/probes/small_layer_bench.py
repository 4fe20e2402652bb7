"""Developer probe: the decoder layer at 16 / 32 token rows -- crag_enc_small_gemm (four forms) and
crag_enc_small_attention per launch (six weight copies round robin, nothing cache resident), next to
crag_enc_skinny_gemm / the unfused kernels, and the whole 36-layer one-query forward in both versions."""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops
from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16


def timeit(fn, n=240, warm=12):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


forms = [("qkv (norm prologue)", 2560, 6144, 12, False, True), ("o", 4096, 2560, 10, False, False),
         ("gate|up (norm prologue, SwiGLU)", 2560, 19456, 16, True, True), ("down", 9728, 2560, 10, False, False)]
for m in (16, 32):
    tot_new = tot_old = 0.0
    for name, k, n, rows, sw, pro in forms:
        ws = [(torch.randn(n, k, generator=g, device=dev) * 0.02).to(BF) for _ in range(6)]
        new_w = [ops.skinny_gate_up_weight(w) if sw else ops.small_weight(w, rows) for w in ws]
        old_w = [ops.skinny_gate_up_weight(w) if sw else ops.skinny_weight(w) for w in ws]
        x = torch.randn(m, k, generator=g, device=dev).to(BF)
        d = torch.randn(m, k, generator=g, device=dev).to(BF)
        nw = torch.ones(k, device=dev, dtype=BF)
        res = torch.empty_like(x)
        normed = torch.empty_like(x)
        out = torch.empty(m, n // 2 if sw else n, dtype=BF, device=dev)

        def run_new(i):
            if pro:
                ops.small_gemm(x, new_w[i % 6], out, m, n, rows, swiglu=sw, delta=d, norm_w=nw, res_out=res)
            else:
                ops.small_gemm(x, new_w[i % 6], out, m, n, rows)

        def run_old(i):
            if pro:
                ops.rmsnorm(d, nw, normed, 1e-6, residual_in=x, residual_out=res)
            ops.skinny_gemm(normed if pro else x, old_w[i % 6], out, m, n, swiglu=sw)

        tn, to = timeit(run_new), timeit(run_old)
        b = n * k * 2
        tot_new += tn
        tot_old += to
        print(f"M={m:2d} {name:34s} K={k} N={n}: new {tn * 1e6:6.1f} us = {b / tn / 1e9:5.0f} GB/s | "
              f"v1 (+ rmsnorm launch) {to * 1e6:6.1f} us = {b / to / 1e9:5.0f} GB/s", flush=True)
        del ws, new_w, old_w
    # attention: 32 q heads / 8 kv heads
    hq, hkv = 32, 8
    qkv = torch.randn(m + 32, (hq + 2 * hkv) * 128, generator=g, device=dev).to(BF)
    qw = torch.ones(128, device=dev, dtype=BF)
    table = Qwen3Encoder._rope_table(Qwen3Config(max_length=64)).to(dev)
    batch = PackedBatch.build([m], dev)
    outa = torch.empty(m, hq * 128, dtype=BF, device=dev)
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=dev)
    qk2 = qkv.clone()

    def attn_new(i):
        ops.small_attention(qkv[:m], qw, qw, table, batch.positions, outa, hq, hkv, 1e-6, 1 / math.sqrt(128))

    def attn_old(i):
        ops.qk_rope_vt(qk2, qw, qw, table, batch.positions, hq, hkv, 1e-6, vt, batch.tok_of_pad)
        ops.attention(qk2, vt, outa, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))

    tn, to = timeit(attn_new), timeit(attn_old)
    print(f"M={m:2d} attention: new {tn * 1e6:6.1f} us | v1 (rope+vt launch, flash attention launch) {to * 1e6:6.1f} us", flush=True)
    print(f"M={m}: per layer new {(tot_new + tn) * 1e6:.1f} us, v1 {(tot_old + to) * 1e6:.1f} us "
          f"(back-to-back launches, host-paced; 202 MB of weights: 25 us at 8 TB/s)", flush=True)

enc = Qwen3Encoder.random_init(Qwen3Config(), seed=1, device=dev)
rng = np.random.default_rng(0)
for v1 in (False, True):
    if v1:
        os.environ["CRAG_ENC_SMALL_V1"] = "1"
    else:
        os.environ.pop("CRAG_ENC_SMALL_V1", None)
    for nq, ntok in ((1, 16), (1, 32), (2, 16)):
        toks = [rng.integers(0, 150000, size=ntok).tolist() for _ in range(nq)]
        for _ in range(3):
            enc.embed_token_lists(toks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            enc.embed_token_lists(toks)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{'v1 (8 launches/layer)' if v1 else 'five-launch layer'}: {nq} x {ntok} tokens: {dt * 1e3:.3f} ms per forward "
              f"({36 * 202.4e6 / dt / 1e12:.2f} TB/s of weights)", flush=True)
