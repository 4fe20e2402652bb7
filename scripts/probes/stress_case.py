"""Developer probe: replay one case of tests/stress_search.py (SEED, CASE) and print where the GPU answer and the
oracle differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle
from cadence_rag_amd.dense_index import DenseIndex
from helpers import random_search_case

seed, target = int(os.environ.get("SEED", 7)), int(os.environ.get("CASE", 78))
rng = np.random.default_rng(seed)
for case in range(target + 1):
    c = random_search_case(rng)
n, nq, k, dim, mask_p, corpus, q, mask = (c[x] for x in ("n", "nq", "k", "dim", "mask_p", "corpus", "queries", "mask"))
print(f"case {target}: n={n} nq={nq} k={k} dim={dim} mask={mask_p}")
ix = DenseIndex(dim, capacity=n)
ix.add(corpus)
ids, sc, ct = ix.search(q, k, row_mask=None if mask is None else DenseIndex.pack_mask(mask))
print("kernel", ix.last_scan_kernel())
wi, ws, wc = oracle.exact_topk(q, corpus, k, mask=None if mask is None else np.packbits(mask, axis=-1, bitorder="little"), mode=oracle.F64)
for qi in range(nq):
    if not np.array_equal(ids[qi], wi[qi]):
        bad = np.nonzero(ids[qi] != wi[qi])[0]
        print(f"query {qi}: counts {ct[qi]} vs {wc[qi]}; differing positions {bad.tolist()}")
        for p_ in bad[:6]:
            print(f"  pos {p_}: gpu id {ids[qi, p_]} score {sc[qi, p_]!r} | oracle id {wi[qi, p_]} score {ws[qi, p_]!r}")
        lo = max(0, int(bad[0]) - 2)
        print("  gpu   ", ids[qi, lo:lo + 6].tolist(), sc[qi, lo:lo + 6].tolist())
        print("  oracle", wi[qi, lo:lo + 6].tolist(), ws[qi, lo:lo + 6].tolist())
        # exact scores of the two contested rows in fp64
        for r in (int(ids[qi, bad[0]]), int(wi[qi, bad[0]])):
            c = corpus[r].astype(np.float64); qq = q[qi].astype(np.float64)
            print(f"  row {r}: fp64 cosine {float(c @ qq / np.linalg.norm(c) / np.linalg.norm(qq))!r}")
ix.close()
