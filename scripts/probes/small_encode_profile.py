"""Developer probe: the 36-layer encoder on ONE short query (the reference's /retrieve operating point), graph replay
and eager; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder

dev = torch.device("cuda", 0)
enc = Qwen3Encoder.random_init(Qwen3Config(), seed=1, device=dev)
rng = np.random.default_rng(0)
for nq, ntok in ((1, 16), (8, 16), (64, 16)):
    toks = [rng.integers(0, 150000, size=ntok).tolist() for _ in range(nq)]
    for mode in ("graph", "eager"):
        if mode == "eager":
            os.environ["CRAG_ENC_NO_GRAPH"] = "1"
        else:
            os.environ.pop("CRAG_ENC_NO_GRAPH", None)
        for _ in range(3):
            enc.embed_token_lists(toks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            enc.embed_token_lists(toks)
            torch.cuda.synchronize()
        print(f"nq={nq} tokens={ntok} {mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward", flush=True)
