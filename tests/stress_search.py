"""Developer stress run (GPU box): many random shapes / k / masks against the fp64 oracle.  Not collected by pytest."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle
from cadence_rag_amd.dense_index import DenseIndex
from helpers import assert_topk_matches, random_search_case

n_cases = int(os.environ.get("CASES", 300))
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
t0 = time.time()
fails = 0
for case in range(n_cases):
    c = random_search_case(rng)
    n, nq, k, dim, mask_p, corpus, q, mask = (c[x] for x in ("n", "nq", "k", "dim", "mask_p", "corpus", "queries", "mask"))
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
