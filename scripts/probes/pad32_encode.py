"""Developer probe: one query of 17-32 tokens (and two of 16) as 32 rows through the five-launch layer against the same
batch padded to 64 rows (the default; CRAG_ENC_NO_PAD_32=1 keeps 32 rows: wide gate|up / down, library qkv / o, one attention launch)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
dev = torch.device("cuda", 0)
cfg = Qwen3Config()
enc = Qwen3Encoder.random_init(cfg, seed=1234, device=dev)
rng = np.random.default_rng(5)
def lat(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best
for nq, ntok in ((1, 24), (1, 32), (2, 16)):
    toks = [rng.integers(0, cfg.vocab_size, size=ntok).tolist() for _ in range(nq)]
    row = f"{nq} x {ntok} tokens:"
    for pad in (False, True):
        if pad: os.environ.pop("CRAG_ENC_NO_PAD_32", None)
        else: os.environ["CRAG_ENC_NO_PAD_32"] = "1"
        row += f"  {'as 64 rows' if pad else 'as 32 rows'} {lat(lambda: enc.embed_token_lists(toks)) * 1e3:6.3f} ms"
    print(row, flush=True)
