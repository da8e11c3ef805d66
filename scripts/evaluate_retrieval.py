#!/usr/bin/env python3
"""Retrieval evaluation — counterpart of the reference's scripts/evaluate_retrieval.py
(metrics :30-62, evaluate_one :65-125, summary :240-260) on the MI355X engine.
The committed reference script crashes on every query (uses math.log2 without
importing math, :49); the metric code here lives in legal_rag_amd/evaluation.py.

    python scripts/evaluate_retrieval.py --data-dir data --lang en --eval-path data/eval/law_qa.jsonl
    python scripts/evaluate_retrieval.py --data-dir data --lang en --synthetic     # seeded offline query set

Each eval line: {"query": str, "article_id": str}.  Per query the channels are
fetched top_k*8 deep, fused, the graph channel is run over the fused seeds
("fused+graph", :97-99; empty when no graph file exists) and the full search()
is run, exactly as the reference does.  The reference routes every query through
its QueryRouter first (:78); routing is outside this path, so the graph walk runs
with decision=None (the walk itself does not read the decision)."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402

from legal_rag_amd import evaluation  # noqa: E402
from legal_rag_amd.config import AppConfig  # noqa: E402
from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever  # noqa: E402


def evaluate_one(query, positives, retriever, top_k, seed_k=None):
    gold = set(map(str.strip, positives))
    eff = top_k * 8
    dense = retriever.search_dense(query, eff)
    bm25 = retriever.search_bm25(query, eff)
    colbert = retriever.search_colbert(query, eff)
    fused = retriever._fuse(dense_hits=dense, bm25_hits=bm25, colbert_hits=colbert)
    seeds = fused[: (seed_k or max(10, top_k * 3))]
    fused_graph = seeds + retriever.search_graph(query, eff, decision=None, seeds=seeds)
    hybrid = retriever.search(query, top_k=eff)
    systems = {"bm25": bm25, "dense": dense, "colbert": colbert, "fused": fused, "fused+graph": fused_graph,
               "hybrid": hybrid}
    return {name: evaluation.all_metrics(evaluation.get_hit_ids(h), gold) for name, h in systems.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--lang", default="en")
    ap.add_argument("--eval-path", type=Path, default=Path("data/eval/law_qa.jsonl"))
    ap.add_argument("--synthetic", action="store_true", help="use the seeded title/span query set built from the index")
    ap.add_argument("--systems", default="bm25,dense,colbert,fused,fused+graph,hybrid")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--encoder-backend", default="auto")
    ap.add_argument("--no-rerank", action="store_true")
    ap.add_argument("--output", type=Path, default=None)
    a = ap.parse_args(argv)

    cfg = AppConfig.for_data_dir(a.data_dir, a.lang)
    cfg.retrieval.encoder_backend = a.encoder_backend
    if a.no_rerank:
        cfg.retrieval.enable_rerank = False
    retriever = HybridRetriever(cfg)
    if a.synthetic:
        retriever.dense.store.load()
        items = [{"query": q, "article_id": g} for q, g, _ in evaluation.synthetic_queries(retriever.dense.store.chunks)]
    else:
        if not a.eval_path.exists():
            raise SystemExit(f"Evaluation file not found: {a.eval_path}")
        items = [json.loads(l) for l in a.eval_path.read_text(encoding="utf-8").splitlines() if l.strip()]
    if a.limit:
        items = items[: a.limit]
    want = [s.strip() for s in a.systems.split(",")]
    rows = []
    for it in items:
        m = evaluate_one(it["query"], [it["article_id"]], retriever, a.top_k)
        for name in want:
            if name in m:
                rows.append({"query": it["query"], "system": name, **m[name]})
    keys = ["R@5", "R@10", "MRR@10", "nDCG@10", "Hit@3", "Hit@10"]
    print(f"\nEvaluation Summary over {len(items)} queries (mean / std):")
    print(f"{'system':12s} " + " ".join(f"{k:>14s}" for k in keys))
    for name in want:
        sel = [r for r in rows if r["system"] == name]
        if sel:
            print(f"{name:12s} " + " ".join(f"{np.mean([r[k] for r in sel]):6.3f}/{np.std([r[k] for r in sel]):6.3f} "
                                            for k in keys))
    if a.output:
        with a.output.open("w", encoding="utf-8") as f:
            for r in rows:
                f.write(json.dumps(r, ensure_ascii=False) + "\n")


if __name__ == "__main__":
    main()
