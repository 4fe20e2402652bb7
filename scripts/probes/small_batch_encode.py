"""Developer probe: encode latency of 1 / 2 / 4 / 8 / 64 queries of 16 tokens (graph replays; the gateway's batch sizes,
RUNBOOK:304,331-334) through the 36-layer 4B architecture, with and without the wide projections (CRAG_ENC_NO_WIDE)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
dev = torch.device("cuda", 0)
cfg = Qwen3Config()
enc = Qwen3Encoder.random_init(cfg, seed=1234, device=dev)
rng = np.random.default_rng(5)
weight_bytes = 2 * cfg.num_layers * (cfg.hidden_size * (cfg.q_size + 2 * cfg.kv_size) + cfg.q_size * cfg.hidden_size + 3 * cfg.hidden_size * cfg.intermediate_size)
def lat(fn, n=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for nq, ntok in ((1, 16), (1, 24), (1, 32), (2, 16), (3, 16), (4, 16), (5, 16), (6, 16), (7, 16), (8, 16), (64, 16)):
    toks = [rng.integers(0, cfg.vocab_size, size=ntok).tolist() for _ in range(nq)]
    row = f"nq {nq:2d} x {ntok} tokens:"
    for wide in (True, False):
        if wide: os.environ.pop("CRAG_ENC_NO_WIDE", None)
        else: os.environ["CRAG_ENC_NO_WIDE"] = "1"
        ms = lat(lambda: enc.embed_token_lists(toks)) * 1e3
        row += f"  {'wide' if wide else 'library'} {ms:6.3f} ms ({weight_bytes / (ms * 1e-3) / 8e12:.3f} of the weight stream)"
    print(row, flush=True)
