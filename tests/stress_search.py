"""Developer stress run (GPU box): many random shapes / k / masks against the fp64 oracle.  Not collected by pytest."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle
from cadence_rag_amd.dense_index import DenseIndex
from helpers import assert_topk_matches

n_cases = int(os.environ.get("CASES", 300))
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
t0 = time.time()
fails = 0
for case in range(n_cases):
    n = int(rng.choice([1, 31, 33, 200, 777, 2500, 6000, 20000, 40000, 66000]))  # the last two: prefilter path
    nq = int(rng.integers(1, 71))
    k = int(rng.choice([1, 5, 10, 31, 32, 33, 50, 64, 65, 100, 128]))
    dim = int(rng.choice([1024, 1024, 1024, 1024, 260, 7]))
    mask_p = rng.choice([-1, -1, 0.0, 0.01, 0.3, 1.0])
    corpus = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 10:
        corpus[n - 1] = corpus[2]
        if rng.random() < 0.3:
            corpus[rng.integers(0, n, size=3)] = 0.0
        if rng.random() < 0.2:
            corpus[5:9] = corpus[5]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    mask = None if mask_p < 0 else (rng.random((nq, n)) < mask_p)
    ix = DenseIndex(dim, capacity=n)
    try:
        ix.add(corpus)
        got = ix.search(q, k, row_mask=None if mask is None else DenseIndex.pack_mask(mask))
        want = oracle.exact_topk(q, corpus, k, mask=None if mask is None else np.packbits(mask, axis=-1, bitorder="little"),
                                 mode=oracle.F64)
        assert_topk_matches(*got, *want, tol=1e-4)
    except AssertionError as exc:
        fails += 1
        print(f"FAIL case {case}: n={n} nq={nq} k={k} dim={dim} mask={mask_p}: {str(exc)[:200]}", flush=True)
    finally:
        ix.close()
    if case % 50 == 49:
        print(f"{case + 1} cases, {fails} failures, {time.time() - t0:.0f}s", flush=True)
print("STRESS", "OK" if fails == 0 else f"{fails} FAILURES")
