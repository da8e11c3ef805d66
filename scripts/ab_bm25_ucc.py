#!/usr/bin/env python3
"""A/B of the BM25 ranking paths on a bench batch (GPU box, one process, interleaved):
AMDR_BM25_SELECT=1 (fp32-image candidates + exact check) vs 0 (exact fp64 arg-max rounds).
    python scripts/ab_bm25_ucc.py [repeat] [en|zh]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_native.load()
W = bench.build_corpus(sys.argv[2] if len(sys.argv) > 2 else "en")
R = bench.Resident(torch, W, 0, rep=rep)
K = 10
R.reserve(K)
out = {}
modes = [("fp32-image select + check", {"AMDR_BM25_SELECT": "1"}), ("exact fp64 arg-max rounds", {"AMDR_BM25_SELECT": "0"})]
times = {m[0]: [] for m in modes}
for r in range(8):
    for name, env in modes:
        os.environ.update(env)
        R.eng.bm25_topk(R.q_terms, R.q_ptr, K)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            s, i = R.eng.bm25_topk(R.q_terms, R.q_ptr, K)
        torch.cuda.synchronize()
        if r:
            times[name].append((time.perf_counter() - t) / 10 * 1e6)
        out[name] = (s.clone(), i.clone())
ref = out["exact fp64 arg-max rounds"]
for name, _ in modes:
    t = sorted(times[name])
    same = bool(torch.equal(out[name][0], ref[0]) and torch.equal(out[name][1], ref[1]))
    print(f"{name:28s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us per {R.nq} queries  identical={same}")
