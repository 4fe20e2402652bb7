"""A seeded slice of the developer stress runs (tests/stress_search.py: 600 cases, tests/stress_small_encode.py: 400)
inside `-m gpu`, so that the driver's own run carries them: random shapes around the kernels' switch points, k up to
128, duplicates, zero rows, per-query masks against the fp64 oracle; short token lists through the graph-replayed
weight-streaming layer against the eager packed forward."""
import os

import numpy as np
import pytest

import oracle
from cadence_rag_amd.dense_index import DenseIndex
from tests.helpers import assert_topk_matches, random_search_case, random_short_token_lists

pytestmark = pytest.mark.gpu


def test_forty_random_search_cases_against_the_fp64_oracle(gpu):
    rng = np.random.default_rng(20261005)
    kernels = set()
    for case in range(40):
        c = random_search_case(rng)
        mask = c["mask"]
        with DenseIndex(c["dim"], capacity=c["n"]) as ix:
            ix.add(c["corpus"])
            got = ix.search(c["queries"], c["k"], row_mask=None if mask is None else DenseIndex.pack_mask(mask))
            kernels.add(ix.last_scan_kernel().split("<")[0])
        want = oracle.exact_topk(c["queries"], c["corpus"], c["k"],
                                 mask=None if mask is None else np.packbits(mask, axis=-1, bitorder="little"),
                                 mode=oracle.F64)
        try:
            assert_topk_matches(*got, *want, tol=1e-4)
        except AssertionError as exc:
            raise AssertionError(f"case {case}: n={c['n']} nq={c['nq']} k={c['k']} dim={c['dim']} "
                                 f"mask={c['mask_p']}: {exc}") from exc
    assert any("prefilter" in name for name in kernels) and any("scan" in name for name in kernels), kernels


def test_twenty_random_short_encodes_replay_equals_eager(gpu, monkeypatch):
    import torch
    from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
    rng = np.random.default_rng(3)
    cfg = Qwen3Config(num_layers=4, vocab_size=4096)
    enc = Qwen3Encoder.random_init(cfg, seed=11, device=torch.device("cuda", 0))
    for case in range(28):
        # 20 cases of the 16 / 32-row shapes (the five-launch layer), 8 of the 64 / 128-row shapes (the gateway's batches
        # of 3 to 8 queries: gate|up and down through the wide weight-streaming kernels)
        lens, toks = random_short_token_lists(rng, cfg.vocab_size) if case < 20 else \
            random_short_token_lists(rng, cfg.vocab_size, shapes=("4x16", "8x16", "1x64", "1x128", "2x64", "4x32"))
        monkeypatch.delenv("CRAG_ENC_NO_GRAPH", raising=False)
        monkeypatch.delenv("CRAG_ENC_NO_SKINNY", raising=False)
        fast = enc.embed_token_lists(toks)
        again = enc.embed_token_lists(toks)
        monkeypatch.setenv("CRAG_ENC_NO_GRAPH", "1")
        monkeypatch.setenv("CRAG_ENC_NO_SKINNY", "1")
        eager = enc.embed_token_lists(toks)
        d = float((fast - eager).abs().max())
        cos = float((fast * eager).sum(-1).min())
        assert torch.equal(fast, again), f"case {case}: lens={lens}: a replay differs from the replay before it"
        assert bool(torch.isfinite(fast).all()) and d < 4e-3 and cos > 0.9997, f"case {case}: lens={lens} d={d} cos={cos}"
