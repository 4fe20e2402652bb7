"""The hardware assumption behind PF_DELTA (crag_search.hip): the conversion to fp16 (v_cvt_pk_f16_f32) and the fp16
MFMA (v_mfma_f32_32x32x16_f16) keep fp16 SUBNORMALS (gradual underflow, |error| <= 2^-25 per element).  A part that
flushed them would be off by up to sqrt(dim) * 2^-14 = 1.95e-3 > PF_DELTA on the vectors of tests/subnormal_cases.py
and drop true neighbours silently.

CPU half: the inputs have teeth (a flushing model of the candidate rule loses the planted neighbours, the
gradual-underflow model keeps them).  GPU half: the real prefilter path == the plain fp32 scan of the same library bit
for bit == the fp64 oracle on those inputs.
"""
import numpy as np
import pytest

import oracle
from cadence_rag_amd.dense_index import DenseIndex
from tests.helpers import assert_topk_matches
from tests.subnormal_cases import build_case, simulate_prefilter

PF_DELTA = 1.25e-3   # crag_search.hip
CASES = [dict(), dict(mixed_signs=True), dict(query_side=True), dict(query_side=True, mixed_signs=True),
         dict(dim=768, mixed_signs=True), dict(dim=768, query_side=True)]
IDS = ["rows", "rows-mixed-signs", "queries", "queries-mixed-signs", "dim768-rows", "dim768-queries"]


@pytest.mark.parametrize("kw", CASES[:2] + CASES[3:5], ids=IDS[:2] + IDS[3:5])
def test_inputs_would_expose_a_flushing_implementation(kw):
    corpus, q, victims = build_case(11, n=6000, **kw)
    kept, top = simulate_prefilter(corpus, q[0], 10, PF_DELTA, flush=False)
    assert set(victims.tolist()) <= set(top.tolist())              # the planted rows ARE true neighbours
    assert set(top.tolist()) <= set(kept.tolist())                 # gradual underflow: the rule keeps all of them
    kept, top = simulate_prefilter(corpus, q[0], 10, PF_DELTA, flush=True)
    assert set(victims.tolist()).isdisjoint(kept.tolist())         # flushed subnormals: every planted neighbour is lost


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CASES, ids=IDS)
def test_prefilter_keeps_neighbours_whose_mass_is_fp16_subnormal(gpu, monkeypatch, kw):
    corpus, q, victims = build_case(23, n=40_000, **kw)
    dim = corpus.shape[1]
    want = oracle.exact_topk(q, corpus, 10, mode=oracle.F64, fast=True)
    assert set(victims.tolist()) <= set(want[0][0].tolist())
    out = {}
    for mirror in (True, False):      # both operand sources of the fp16 scan: the stored mirror, the on-the-fly conversion
        if mirror:
            monkeypatch.delenv("CRAG_NO_FP16_MIRROR", raising=False)
        else:
            monkeypatch.setenv("CRAG_NO_FP16_MIRROR", "1")
        for pf in (True, False):
            if pf:
                monkeypatch.delenv("CRAG_NO_PREFILTER", raising=False)
            else:
                monkeypatch.setenv("CRAG_NO_PREFILTER", "1")
            with DenseIndex(dim, capacity=len(corpus)) as ix:
                ix.add(corpus)
                res = [ix.search(q, k) for k in (10, 10, 50)]   # forward pass, reversed pass, a k on two class sets
                assert ("prefilter" in ix.last_scan_kernel()) == pf, ix.last_scan_kernel()
            out[(mirror, pf)] = res
    ref = out[(True, False)]
    for key, res in out.items():
        for a, b in zip(res, ref):
            for x, y in zip(a, b):
                assert np.array_equal(x, y, equal_nan=True), f"mirror={key[0]} prefilter={key[1]} differs from the fp32 scan"
    got = out[(True, True)][0]
    assert set(victims.tolist()) <= set(got[0][0].tolist())
    assert_topk_matches(*got, *want, tol=1e-4)
