"""Developer probe: ONE 16-token query (or argv[1] queries) through the 36-layer encoder, graph replay only, for
`rocprofv3 --kernel-trace --stats` (per-kernel durations of the five-launch layer) -- and with CRAG_TRACE_GAPS=1 the
start/end stamps of one replay's kernels from the trace csv are summarised by trace_gaps.py."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder

dev = torch.device("cuda", 0)
enc = Qwen3Encoder.random_init(Qwen3Config(), seed=1, device=dev)
rng = np.random.default_rng(0)
ntok = int(os.environ.get("NTOK", "16"))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # queries per call
toks = [rng.integers(0, 150000, size=ntok).tolist() for _ in range(nq)]
for _ in range(3):
    enc.embed_token_lists(toks)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    enc.embed_token_lists(toks)
    torch.cuda.synchronize()
print(f"{nq} x {ntok} tokens: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward", flush=True)
