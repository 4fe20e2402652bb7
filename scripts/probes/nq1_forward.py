"""Developer probe: the one-query forward (1 x 16 tokens, graph replay) -- prints ms per forward; run once per
kernel variant under test (e.g. CRAG_SMALL_DOWN_VARIANT=1 python scripts/probes/nq1_forward.py)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder

dev = torch.device("cuda", 0)
enc = Qwen3Encoder.random_init(Qwen3Config(), seed=1, device=dev)
rng = np.random.default_rng(0)
toks = [rng.integers(0, 150000, size=16).tolist()]
for _ in range(5):
    enc.embed_token_lists(toks)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        enc.embed_token_lists(toks)
        torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CRAG_SMALL_"))
print(f"[{tag or 'default'}] 1 x 16 tokens: {best * 1e3:.3f} ms per forward ({36 * 202.4e6 / best / 1e12:.2f} TB/s of weights)", flush=True)
