"""Adversarial inputs for the fp16 prefilter's error bound (crag_search.hip: PF_DELTA): unit vectors whose mass,
apart from one or two large components, sits in fp16's SUBNORMAL range (|x| < 2^-14 after normalisation).

If the conversion to fp16 or the fp16 MFMA flushed subnormals to zero, the approximate cosine of such a vector against
a query proportional to a sign pattern would be off by ||small part|| ~ sqrt(dim) * 2^-14 = 1.95e-3 at dim 1024 — more
than PF_DELTA = 1.25e-3 — and with the error pointing one way for the true neighbours ("victims": small components
aligned with the query) and the other way for the rows ranked just below them ("competitors": small components
against the query) the gap of 3.9e-3 exceeds the 2*delta window the candidate rule keeps: true neighbours would be
dropped silently.  With gradual underflow (what the bound's proof assumes) every element is rounded by at most 2^-25
and nothing is lost.

`simulate_prefilter` restates the candidate rule on the CPU for both behaviours, so the `not gpu` suite proves that
these inputs would catch a flushing implementation; the `-m gpu` suite runs them through the real kernels.
"""
from __future__ import annotations

import numpy as np

F16_MIN_NORMAL = 2.0 ** -14


def _signs(rng, dim, mixed):
    return rng.choice(np.array([-1.0, 1.0]), size=dim) if mixed else np.ones(dim)


def build_case(seed: int, n: int = 40_000, dim: int = 1024, mixed_signs: bool = False, query_side: bool = False,
               n_victims: int = 6, n_comp: int = 600):
    """Returns (corpus [n, dim] f32, queries [nq, dim] f32, victims: row positions that are in the exact top-10 of
    query 0 and whose approximate score a flushing implementation under-estimates by ~sqrt(dim) * 2^-14).

    row side (query_side=False): query 0 = sign pattern / sqrt(dim); victims = one component 1 + subnormal rest aligned
    with the pattern; competitors = two normal components + subnormal rest AGAINST the pattern, exact scores spread
    just below the victims'.
    query side (query_side=True): the roles of the operands are swapped — query 0 has the subnormal mass, the planted
    rows are (anti-)aligned sign patterns with one tuned component."""
    rng = np.random.default_rng(seed)
    sg = _signs(rng, dim, mixed_signs)
    p = int(rng.integers(0, dim))
    u = sg / np.sqrt(dim)
    pos = rng.permutation(n)[: n_victims + n_comp]
    vpos, cpos = pos[:n_victims], pos[n_victims:]

    def small(m):  # magnitudes at the top of fp16's subnormal range
        return rng.uniform(0.85, 0.98, size=(m, dim)) * F16_MIN_NORMAL

    def background(q0):  # unit rows orthogonal to query 0: their scores against it are ~0, far below the planted rows
        bg = rng.standard_normal((n, dim))
        bg -= np.outer(bg @ q0, q0)
        return bg / np.linalg.norm(bg, axis=1, keepdims=True)

    if not query_side:
        q0 = u.copy()
        corpus = background(q0)
        v = small(n_victims) * sg
        v[:, p] = sg[p]
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        s_v = float(np.min(v @ q0))
        # competitors: cos(t) at p and sin(t) at p2 (normal range), subnormal rest against the pattern; the angle is
        # solved per row so that the exact scores fill [s_v - 7e-4, s_v - 2e-5]
        p2 = (p + 1 + int(rng.integers(0, dim - 1))) % dim
        c = -small(n_comp) * sg
        target = s_v - rng.uniform(2e-5, 7e-4, size=n_comp)
        base = c.copy()
        base[:, [p, p2]] = 0.0
        lo, hi = np.zeros(n_comp), np.full(n_comp, np.pi / 4)
        for _ in range(60):   # bisection on the angle, all rows at once (the score grows with t on [0, pi/4])
            t = 0.5 * (lo + hi)
            dot = base @ q0 + np.cos(t) * sg[p] * q0[p] + np.sin(t) * sg[p2] * q0[p2]
            sc = dot / np.sqrt((base * base).sum(axis=1) + 1.0)
            below = sc < target
            lo, hi = np.where(below, t, lo), np.where(below, hi, t)
        c[:, p], c[:, p2] = np.cos(t) * sg[p], np.sin(t) * sg[p2]
        corpus[vpos] = v
        corpus[cpos] = c / np.linalg.norm(c, axis=1, keepdims=True)
        queries = [q0, -q0]
    else:
        q0 = (small(1)[0]) * sg
        q0[p] = sg[p]
        q0 /= np.linalg.norm(q0)
        corpus = background(q0)
        # victims: the aligned pattern (score = q0_p u_p + sum small/sqrt(dim)); competitors: anti-aligned rest with a
        # larger tuned component at p
        v = np.tile(u, (n_victims, 1)) * (1.0 + 1e-3 * rng.standard_normal((n_victims, 1)))
        v += 1e-4 * rng.standard_normal((n_victims, dim)) / np.sqrt(dim)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        s_v = float(np.min(v @ q0))
        c = np.tile(-u, (n_comp, 1))
        target = s_v - rng.uniform(2e-5, 7e-4, size=n_comp)
        base = c.copy()
        base[:, p] = 0.0
        lo, hi = np.zeros(n_comp), np.full(n_comp, 0.5)
        for _ in range(60):   # bisection on the component at p, all rows at once
            x = 0.5 * (lo + hi)
            sc = (base @ q0 + x * sg[p] * q0[p]) / np.sqrt((base * base).sum(axis=1) + x * x)
            below = sc < target
            lo, hi = np.where(below, x, lo), np.where(below, hi, x)
        c[:, p] = x * sg[p]
        corpus[vpos] = v
        corpus[cpos] = c / np.linalg.norm(c, axis=1, keepdims=True)
        queries = [q0, -q0]
    # more queries: perturbed copies (bounds form differently), and random ones
    for e in (3e-4, 1e-3):
        queries.append(queries[0] + e * rng.standard_normal(dim) / np.sqrt(dim))
    for _ in range(28):
        queries.append(rng.standard_normal(dim))
    corpus32 = corpus.astype(np.float32)
    queries32 = np.stack(queries).astype(np.float32)
    return corpus32, queries32, np.sort(vpos)


def simulate_prefilter(corpus: np.ndarray, query: np.ndarray, k: int, delta: float, flush: bool):
    """The candidate rule of prefilter_kernel / finalize_kernel with the BEST possible bound (the true k-th largest
    approximate score; the kernels' bounds are lower, i.e. keep more): approximate cosine = sum f16(q_i) f16(c_i) of
    the normalised operands, a row stays when approx >= kth_approx - 2 delta.  flush=True models hardware that
    replaces fp16 subnormals by zero.  Returns (kept row positions, exact top-k positions)."""
    c = corpus.astype(np.float64)
    q = query.astype(np.float64)
    cn = (c / np.linalg.norm(c, axis=1, keepdims=True)).astype(np.float32)
    qn = (q / np.linalg.norm(q)).astype(np.float32)
    c16 = cn.astype(np.float16).astype(np.float64)
    q16 = qn.astype(np.float16).astype(np.float64)
    if flush:
        c16[np.abs(c16) < F16_MIN_NORMAL] = 0.0
        q16[np.abs(q16) < F16_MIN_NORMAL] = 0.0
    approx = c16 @ q16
    exact = cn.astype(np.float64) @ qn.astype(np.float64)
    kth = np.sort(approx)[-k]
    kept = np.nonzero(approx >= kth - 2.0 * delta)[0]
    top = np.argsort(-exact, kind="stable")[:k]
    return kept, top
