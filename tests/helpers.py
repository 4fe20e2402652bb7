"""Shared test helpers (tests may use the oracle; the product never does)."""
from __future__ import annotations

import numpy as np


def unit_rows(rng, n, dim=1024):
    a = rng.standard_normal((n, dim)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    return a


def assert_topk_matches(ids, scores, counts, want_ids, want_scores, want_counts, tol=1e-4, gap=2e-6):
    """Bit-exact ids wherever the oracle's adjacent scores are separated by more than `gap`
    (fp32 vs fp64 rounding can swap closer pairs -- also across the cut at k, where the partner of the swap is not
    in the list); |dscore| <= tol (BASELINE.json) everywhere."""
    assert np.array_equal(counts, want_counts), (counts, want_counts)
    for q in range(ids.shape[0]):
        c = int(want_counts[q])
        assert np.all(ids[q, c:] == -1) and np.all(np.isnan(scores[q, c:]))
        if c == 0:
            continue
        assert np.max(np.abs(scores[q, :c].astype(np.float64) - want_scores[q, :c])) <= tol
        if np.array_equal(ids[q, :c], want_ids[q, :c]):
            continue
        # allow only permutations inside runs of near-tied oracle scores
        ws = want_scores[q, :c]
        run_start = 0
        full = c == ids.shape[1]   # the list is cut at k: the last run may continue below it, out of sight
        for i in range(1, c + 1):
            if i == c or (ws[i - 1] - ws[i]) > gap:
                got, want = ids[q, run_start:i].tolist(), want_ids[q, run_start:i].tolist()
                if i == c and full and sorted(got) != sorted(want):
                    # a row the oracle ranks just below the cut may stand in for a near-tied row above it:
                    # its own score must then be within `gap` of the oracle's last score
                    for pos in range(run_start, c):
                        if ids[q, pos] not in want:
                            assert float(scores[q, pos]) >= ws[c - 1] - gap, \
                                f"query {q}: id {ids[q, pos]} at the cut is not a near tie of the oracle's last row"
                    got = [g for g in got if g in want]
                    assert len(set(got)) == len(got)
                else:
                    assert sorted(got) == sorted(want), \
                        f"query {q}: ids differ outside a near-tie run at positions {run_start}:{i}"
                run_start = i


def cpu_merge_topk(g_ids, g_sc, g_ct):
    """Reference merge for the gloo tests: [R, nq, k] -> [nq, k], score desc then id asc."""
    import torch
    r, nq, k = g_ids.shape
    out_ids = torch.full((nq, k), -1, dtype=torch.int64)
    out_sc = torch.full((nq, k), float("nan"), dtype=torch.float32)
    out_ct = torch.zeros((nq,), dtype=torch.int32)
    for q in range(nq):
        cand = []
        for s in range(r):
            for j in range(int(g_ct[s, q])):
                cand.append((-float(g_sc[s, q, j]), int(g_ids[s, q, j])))
        cand.sort()
        cand = cand[:k]
        out_ct[q] = len(cand)
        for j, (ns, i) in enumerate(cand):
            out_ids[q, j] = i
            out_sc[q, j] = -ns
    return out_ids, out_sc, out_ct


def random_search_case(rng):
    """One random search case of the stress run (tests/stress_search.py, scripts/probes/stress_case.py and the slice of
    it that -m gpu carries): shapes around the kernels' switch points, duplicates, zero rows, per-query masks.
    The draw order is part of the contract: (SEED, CASE) of a failure replays it."""
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
    return {"n": n, "nq": nq, "k": k, "dim": dim, "mask_p": float(mask_p), "corpus": corpus, "queries": q, "mask": mask}


def random_short_token_lists(rng, vocab_size, shapes=("1x16", "1x32", "2x16")):
    """One case of tests/stress_small_encode.py: token lists whose padded shape is one of the small-graph shapes."""
    shape = rng.choice(list(shapes))
    if shape == "1x16":
        lens = [int(rng.integers(1, 17))]
    elif shape == "1x32":
        lens = [int(rng.integers(17, 33))]
    elif shape == "2x16":
        lens = [int(rng.integers(1, 17)), int(rng.integers(1, 17))]
    else:   # "BxL": B sequences of up to L tokens (at least one reaches the bucket)
        b, l = (int(x) for x in shape.split("x"))
        lens = [int(rng.integers(max(1, l // 2 + 1), l + 1))] + [int(rng.integers(1, l + 1)) for _ in range(b - 1)]
    return lens, [rng.integers(0, vocab_size, size=n).tolist() for n in lens]


def cpu_merge_topk_packed(gathered, world: int, nq: int, k: int, record_bytes: int):
    """CPU stand-in of merge_packed_kernel (crag_merge_topk_packed): `gathered` is the uint8 tensor an all-gather of
    ResultRecords delivers -- per rank `record_bytes` bytes laid out ids int64 [nq, k] | scores fp32 [nq, k] | counts
    int32 [nq] | padding to 8 bytes (include/crag_dense.h: crag_result_record_bytes) -- parsed where it lies."""
    import torch
    raw = gathered.numpy()
    assert raw.dtype == np.uint8 and raw.size == world * record_bytes
    g_ids = np.empty((world, nq, k), dtype=np.int64)
    g_sc = np.empty((world, nq, k), dtype=np.float32)
    g_ct = np.empty((world, nq), dtype=np.int32)
    for r in range(world):
        rec = raw[r * record_bytes:(r + 1) * record_bytes]
        g_ids[r] = rec[:nq * k * 8].view(np.int64).reshape(nq, k)
        g_sc[r] = rec[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k)
        g_ct[r] = rec[nq * k * 12:nq * k * 12 + nq * 4].view(np.int32)
    return cpu_merge_topk(torch.from_numpy(g_ids), torch.from_numpy(g_sc), torch.from_numpy(g_ct))
