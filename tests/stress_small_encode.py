"""Developer stress run (GPU box): random short token lists whose padded shapes are 16 / 32 rows -- the five-launch
layer behind a graph replay (csrc/crag_encoder_small.hip) -- or 64 / 128 rows (the wide projections, crag_encoder_wide.hip) against the eager packed forward through the library GEMMs
and the unfused kernels, at the 4B widths (4 layers).  Not collected by pytest."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
from helpers import random_short_token_lists

n_cases = int(os.environ.get("CASES", 300))
rng = np.random.default_rng(int(os.environ.get("SEED", 3)))
dev = torch.device("cuda", 0)
cfg = Qwen3Config(num_layers=4, vocab_size=4096)
enc = Qwen3Encoder.random_init(cfg, seed=11, device=dev)
worst, fails, t0 = 0.0, 0, time.time()
for case in range(n_cases):
    lens, toks = random_short_token_lists(rng, cfg.vocab_size, shapes=("1x16", "1x32", "2x16", "4x16", "8x16", "1x64", "1x128", "2x64", "4x32"))
    os.environ.pop("CRAG_ENC_NO_GRAPH", None); os.environ.pop("CRAG_ENC_NO_SKINNY", None)
    fast = enc.embed_token_lists(toks)
    again = enc.embed_token_lists(toks)
    os.environ["CRAG_ENC_NO_GRAPH"] = os.environ["CRAG_ENC_NO_SKINNY"] = "1"
    eager = enc.embed_token_lists(toks)
    d = float((fast - eager).abs().max())
    cos = float((fast * eager).sum(-1).min())
    worst = max(worst, d)
    ok = torch.equal(fast, again) and bool(torch.isfinite(fast).all()) and d < 4e-3 and cos > 0.9997
    if not ok:
        fails += 1
        print(f"FAIL case {case}: lens={lens} max|d|={d:.2e} cos={cos:.6f} replay_equal={torch.equal(fast, again)}", flush=True)
    if case % 100 == 99:
        print(f"{case + 1} cases, {fails} failures, worst max|d| {worst:.2e}, {time.time() - t0:.0f}s", flush=True)
print("STRESS", "OK" if fails == 0 else f"{fails} FAILURES", f"(worst max|d| vs the eager forward {worst:.2e})")
