#!/usr/bin/env python3
"""A/B of the BM25 ranking paths on the UCC-en bench batch (GPU box, one process, interleaved):
AMDR_BM25_SELECT=1 (fp32-image candidates + exact check) vs 0 (exact fp64 arg-max rounds).
    python scripts/ab_bm25_ucc.py [repeat]"""
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
W = bench.build_corpus("en")
R = bench.Resident(torch, W, 0, rep=rep)
K = 10
R.reserve(K)
out = {}
times = {"1": [], "0": []}
for r in range(8):
    for flag in ("1", "0"):
        os.environ["AMDR_BM25_SELECT"] = flag
        R.eng.bm25_topk(R.q_terms, R.q_ptr, K)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            s, i = R.eng.bm25_topk(R.q_terms, R.q_ptr, K)
        torch.cuda.synchronize()
        if r:
            times[flag].append((time.perf_counter() - t) / 10 * 1e6)
        out[flag] = (s.clone(), i.clone())
same = bool(torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1]))
for flag, name in (("1", "fp32-image select + exact check"), ("0", "exact fp64 arg-max rounds")):
    t = sorted(times[flag])
    print(f"{name:34s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us per {R.nq} queries")
print("identical results:", same)
