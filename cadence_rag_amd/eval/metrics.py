from __future__ import annotations

import argparse
import json
import math
from typing import Dict, List, Sequence


def load_jsonl(path: str) -> List[dict]:
    with open(path, "r", encoding="utf-8") as fh:
        return [json.loads(line) for line in (ln.strip() for ln in fh) if line]


def dcg(relevances: Sequence[int]) -> float:
    return sum(rel / math.log2(pos + 1) for pos, rel in enumerate(relevances, start=1) if rel > 0)


def compute_metrics(gold: Dict[str, List[str]], results: Dict[str, List[str]], ks: List[int]) -> Dict[str, float]:
    """recall@k = |top-k ∩ relevant| / |relevant|, MRR, binary nDCG@k; queries without relevant ids are
    skipped; averages over the remaining queries (0.0 for everything when none remain)."""
    totals = {f"recall@{k}": 0.0 for k in ks}
    totals["mrr"] = 0.0
    for k in ks:
        totals[f"ndcg@{k}"] = 0.0
    n = 0
    for qid, relevant in gold.items():
        if not relevant:
            continue
        n += 1
        rel = set(relevant)
        got = results.get(qid, [])
        totals["mrr"] += next((1.0 / pos for pos, doc in enumerate(got, start=1) if doc in rel), 0.0)
        for k in ks:
            top = got[:k]
            totals[f"recall@{k}"] += sum(1 for doc in top if doc in rel) / max(len(relevant), 1)
            ideal = dcg([1] * min(len(relevant), k)) or 1.0
            totals[f"ndcg@{k}"] += dcg([1 if doc in rel else 0 for doc in top]) / ideal
    if n == 0:
        return {key: 0.0 for key in totals}
    return {key: val / n for key, val in totals.items()}


def gate_failures(metrics: Dict[str, float], *, min_mrr: float = 0.0, min_recall_at: int = 20, min_recall: float = 0.0,
                  min_ndcg_at: int = 10, min_ndcg: float = 0.0) -> List[str]:
    out = []
    rk, nk = f"recall@{min_recall_at}", f"ndcg@{min_ndcg_at}"
    if metrics.get("mrr", 0.0) < min_mrr:
        out.append(f"mrr {metrics.get('mrr', 0.0):.4f} < {min_mrr:.4f}")
    if metrics.get(rk, 0.0) < min_recall:
        out.append(f"{rk} {metrics.get(rk, 0.0):.4f} < {min_recall:.4f}")
    if metrics.get(nk, 0.0) < min_ndcg:
        out.append(f"{nk} {metrics.get(nk, 0.0):.4f} < {min_ndcg:.4f}")
    return out


def _read(gold_path: str, results_path: str):
    gold = {r["query_id"]: r.get("relevant_ids", []) for r in load_jsonl(gold_path)}
    results = {r["query_id"]: r.get("retrieved_ids", r.get("retrieved", [])) for r in load_jsonl(results_path)}
    return gold, results


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Evaluate retrieval results / fail below thresholds.")
    ap.add_argument("--gold", required=True)
    ap.add_argument("--results", required=True)
    ap.add_argument("--k", nargs="+", type=int, default=[5, 10, 20])
    ap.add_argument("--gate", action="store_true", help="apply the regression-gate thresholds")
    ap.add_argument("--min-mrr", type=float, default=0.0)
    ap.add_argument("--min-recall-at", type=int, default=20)
    ap.add_argument("--min-recall", type=float, default=0.0)
    ap.add_argument("--min-ndcg-at", type=int, default=10)
    ap.add_argument("--min-ndcg", type=float, default=0.0)
    args = ap.parse_args(argv)
    ks = sorted(set(args.k + ([args.min_recall_at, args.min_ndcg_at] if args.gate else [])))
    gold, results = _read(args.gold, args.results)
    metrics = compute_metrics(gold, results, ks if args.gate else args.k)
    print(json.dumps(metrics, indent=2))
    if args.gate:
        failures = gate_failures(metrics, min_mrr=args.min_mrr, min_recall_at=args.min_recall_at,
                                 min_recall=args.min_recall, min_ndcg_at=args.min_ndcg_at, min_ndcg=args.min_ndcg)
        if failures:
            print("[regression_gate] FAIL")
            for f in failures:
                print(f" - {f}")
            raise SystemExit(1)
        print("[regression_gate] PASS")


if __name__ == "__main__":
    main()
