"""Developer probe: one query of 17-32 tokens (and two of 16): 32 rows through the five-launch layer
(CRAG_ENC_NO_WIDE_32=1), the same batch padded to 64 rows (a third phantom-padded query added here), and 32 rows through
the wide gate|up / down kernels + library qkv / o + one attention launch (the default)."""
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
    for mode in ("32 rows, five-launch layer", "as 64 rows", "32 rows, wide kernels"):
        os.environ.pop("CRAG_ENC_NO_WIDE_32", None)
        if mode.startswith("32 rows, five"): os.environ["CRAG_ENC_NO_WIDE_32"] = "1"
        batch = toks
        if mode == "as 64 rows":   # one more one-token query: 33..63 padded rows round up to 64
            batch = toks + [[1]]
        enc.__dict__.pop("_graphs", None)
        row += f"  {mode} {lat(lambda: enc.embed_token_lists(batch)) * 1e3:6.3f} ms"
    print(row, flush=True)
