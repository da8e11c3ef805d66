#!/usr/bin/env python3
"""A/B of the UCC-en hybrid step (GPU box, one process, interleaved):
  serial        dense -> BM25 -> fuse on one stream (the shipped order), persistent dense grid (shipped)
  1 per block   the same with one dense block per logical block (AMDR_PANEL_PERSIST=0)
  side          BM25 on a second, lower-priority stream beside the dense kernels
  side, 1/block the same without the persistent grid (BM25 waves take the LDS a retiring dense block frees)
    python scripts/ab_overlap.py [repeat]
Result kept in DESIGN.md §4.2: beside the persistent dense blocks BM25 is resident on what the dense
waves leave (48 KiB of LDS, 208 VGPRs per SIMD lane) and gets ~1/6 of the vector issue it has alone —
it takes 320-370 us instead of 84, ends after the dense kernel, and the step gains 2-4 %."""
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
params = _native.make_fuse_params(min_final_score=0.2)
eng = R.eng
main = torch.cuda.Stream(priority=-1)
side = torch.cuda.Stream(priority=0)


def step(overlap):
    if overlap:
        side.wait_stream(main)
        d = eng.dense_topk(R.q_emb, K)
        with torch.cuda.stream(side):
            b = eng.bm25_topk(R.q_terms, R.q_ptr, K)
        main.wait_stream(side)
    else:
        d = eng.dense_topk(R.q_emb, K)
        b = eng.bm25_topk(R.q_terms, R.q_ptr, K)
    return eng.fuse(params, R.nq, d, b, None)


modes = [("serial", None, False), ("1 per block", "0", False), ("side", None, True), ("side, 1/block", "0", True)]
times = {m[0]: [] for m in modes}
outs = {}
with torch.cuda.stream(main):
    for r in range(7):
        for name, persist, overlap in modes:
            if persist:
                os.environ["AMDR_PANEL_PERSIST"] = persist
            else:
                os.environ.pop("AMDR_PANEL_PERSIST", None)
            step(overlap)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                res = step(overlap)
            torch.cuda.synchronize()
            if r:
                times[name].append((time.perf_counter() - t) / 20 * 1e6)
            outs[name] = (res.ids.clone(), res.vals.clone(), res.count.clone())
ref = outs["serial"]
for name, _, _ in modes:
    t = sorted(times[name])
    same = all(torch.equal(a, b) for a, b in zip(ref, outs[name]))
    print(f"{name:14s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us per {R.nq}-query step  identical={same}")
