"""Developer probe: attention kernel time vs sequence length at a fixed token count (65 536)."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops
from cadence_rag_amd.encoder.qwen3 import PackedBatch
dev = torch.device("cuda", 0)
hq, hkv = 32, 8
for L in (64, 128, 256, 512, 1024):
    n = 65536 // L
    lens = [L] * n
    t = sum(lens)
    batch = PackedBatch.build(lens, dev)
    qkv = (torch.randn(t + 64, (hq + 2 * hkv) * 128, device=dev) * 0.5).to(torch.bfloat16)
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=torch.bfloat16, device=dev)
    out = torch.empty(t, hq * 128, dtype=torch.bfloat16, device=dev)
    ops.v_transpose(qkv, vt, batch.tok_of_pad, hq, hkv)
    def run():
        ops.attention(qkv, vt, out, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    nb = L // 32
    tiles = n * nb * (nb + 1) // 2          # (q block, key tile) pairs per kv-head group
    flops = tiles * hq * 2 * 2 * 32 * 32 * 128
    print(f"L={L:5d} seqs={n:5d} q-blocks={n*nb:6d} tile-iters/WG={(nb+1)/2:5.1f}: {dt*1e6:8.1f} us  {flops/dt/1e12:7.1f} TF/s (incl. masked half of diagonal tiles)  "
          f"{dt*1e6/(n*nb*hkv)*512:6.2f} us per WG slot", flush=True)
