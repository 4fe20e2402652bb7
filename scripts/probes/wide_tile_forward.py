"""Developer probe: the 32 / 64 / 128-row forward with the wide kernels' tile height forced (CRAG_WIDE_TILE=64 against the
default choice: 128-row tiles for gate|up, 64-row tiles for o / down) -- 1 x 32, 4 x 16 and 8 x 16 tokens."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for nq, ntok in ((1, 32), (4, 16), (8, 16)):
    toks = [rng.integers(0, cfg.vocab_size, size=ntok).tolist() for _ in range(nq)]
    print(f"{nq} x {ntok} tokens [CRAG_WIDE_TILE={os.environ.get('CRAG_WIDE_TILE')} CRAG_ENC_MIX_LAST_ONLY={os.environ.get('CRAG_ENC_MIX_LAST_ONLY')}]: {lat(lambda: enc.embed_token_lists(toks)) * 1e3:.3f} ms", flush=True)
