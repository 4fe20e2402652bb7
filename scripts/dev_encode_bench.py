"""Developer: encode throughput on the full Qwen3-Embedding-4B architecture (random weights)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder, PackedBatch
layers = int(os.environ.get("LAYERS", 36)); B = int(os.environ.get("B", 256))
cfg = Qwen3Config(num_layers=layers)
dev = torch.device("cuda", 0)
t0 = time.time(); enc = Qwen3Encoder.random_init(cfg, seed=0, device=dev); torch.cuda.synchronize(); print("init s", time.time() - t0)
rng = np.random.default_rng(0)
lens = np.clip(rng.normal(256, 96, size=B).round().astype(int), 8, 1024)
lens = (lens * (256 * B / lens.sum())).round().astype(int).clip(8, 1024)
batch = PackedBatch.build(lens, dev)
ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)).to(dev)
for _ in range(2): out = enc.forward_packed(ids, batch)
torch.cuda.synchronize()
n = 5; t0 = time.perf_counter()
for _ in range(n): out = enc.forward_packed(ids, batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
tok = int(lens.sum()); fl = cfg.flops_per_token(float((lens.astype(float) ** 2).sum() / lens.sum())) * tok
print(f"layers={layers} B={B} tokens={tok} step={dt*1e3:.1f}ms chunks/s={B/dt:.1f} tokens/s={tok/dt:.0f} TFLOP/s={fl/dt/1e12:.1f} (scaled to 36 layers: {B/dt*layers/36:.1f} chunks/s)")
print("norms", out.norm(dim=1)[:4].tolist())
