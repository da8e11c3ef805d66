#!/usr/bin/env python3
"""Golden vectors produced by the in-repo oracle.

These pin the ORACLE ITSELF across refactors (regression vectors) and give the
GPU tests fixed known answers; they do not pin the oracle to the third-party
wheels (faiss / rank_bm25 / colbert are absent here: "parity unpinned" for the
arithmetic of those three, see oracle/__init__.py).  The BM25 toy case is small
enough to check by hand: idf / avgdl / scores are printed to 17 digits.

    python tests/golden/gen_oracle_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import bm25 as OB  # noqa: E402
from oracle import dense as OD  # noqa: E402
from oracle import maxsim as OM  # noqa: E402

OUT = Path(__file__).resolve().parent


def unit_rows(rng, n, d):
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X


def main():
    # (3) FlatIP: rng(0) X[4096,768], Q[16,768]; scores from an fp64 product so the
    # fixture does not depend on the BLAS summation order of the generating host
    rng = np.random.default_rng(0)
    X = unit_rows(rng, 4096, 768)
    Q = unit_rows(rng, 16, 768)
    s64 = Q.astype(np.float64) @ X.astype(np.float64).T
    s, i = OD.topk_desc(s64, 10)
    np.savez_compressed(OUT / "dense_flatip_golden.npz", ids=i, scores=s.astype(np.float32))

    # (2) BM25 toy corpus, hand-checkable
    docs = [
        "A contract for the sale of goods may be made in any manner sufficient to show agreement.",
        "The seller's warranty: goods shall be merchantable, and the seller is a merchant of goods of that kind.",
        "An offer by a merchant to buy or sell goods in a signed writing is not revocable.",
        "Unless otherwise agreed, the buyer must pay at the time and place of delivery of the goods.",
        "A security interest attaches to collateral when it becomes enforceable against the debtor.",
        "goods goods goods",
        "",
        "The buyer's remedies: the buyer may cover, and recover damages for non-delivery of goods.",
    ]
    toks = [OB.tokenize_en(t) for t in docs]
    bm = OB.BM25Okapi(toks)
    queries = [["goods"], ["merchant", "goods", "warranty"], ["buyer", "buyer", "pay"], ["zzz"], [],
               ["the", "a", "of"], ["What", "is", "the", "seller's", "warranty", "?"]]
    out = {"docs": docs, "avgdl": bm.avgdl, "idf": bm.idf, "doc_len": bm.doc_len, "queries": []}
    for q in queries:
        sc = bm.get_scores(q)
        order = [i for i, _ in OB.search(bm, q, len(docs))]
        out["queries"].append({"tokens": q, "scores": [float(x) for x in sc], "order": order})
        print(q, [f"{x:.17g}" for x in sc])
    (OUT / "bm25_toy.json").write_text(json.dumps(out, indent=1))

    # (4) MaxSim: seeded Q[2,32,128], ragged docs 1..220
    rng = np.random.default_rng(42)
    lens = rng.integers(1, 221, size=64)
    lens[0], lens[-1] = 1, 220
    doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    D = unit_rows(rng, int(doc_ptr[-1]), 128)
    Qm = unit_rows(rng, 64, 128).reshape(2, 32, 128)
    sc = OM.maxsim_scores(Qm, D, doc_ptr)
    np.savez_compressed(OUT / "maxsim_golden.npz", lens=lens, scores=sc)
    print("wrote goldens")


if __name__ == "__main__":
    main()
