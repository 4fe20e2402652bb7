"""Developer probe: does running two half-batches of the encoder forward on two HIP streams hide the HBM-bound passes
(SwiGLU, RMSNorm, RoPE: 18 % of a layer) behind the other half's MFMA-bound GEMMs?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder

dev = torch.device("cuda", 0)
cfg = Qwen3Config()
enc = Qwen3Encoder.random_init(cfg, seed=1234, device=dev)
rng = np.random.default_rng(2024)
n = 256
lens = np.clip(rng.normal(256, 96, size=n).round().astype(int), 8, 1024)
lens = (lens * (256 * n / lens.sum())).round().astype(int).clip(8, 1024)

def make(ls):
    return PackedBatch.build(ls, dev), torch.from_numpy(rng.integers(0, cfg.vocab_size, size=int(ls.sum())).astype(np.int32)).to(dev)

full = make(lens)
for parts in (1, 2, 3, 4):
    cuts = np.array_split(np.arange(n), parts)
    subs = [make(lens[c]) for c in cuts]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for (b, ids), st in zip(subs, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(enc.forward_packed(ids, b))
        for st in streams:
            cur.wait_stream(st)
        return outs
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 6
    for _ in range(reps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{parts} stream(s): {dt*1e3:7.1f} ms per 256-chunk batch = {n/dt:7.1f} chunks/s", flush=True)
