#!/usr/bin/env python3
"""Same-process A/B of the MaxSim forms on the UCC-en / Civil-Code-zh token stores:
AMDR_MAXSIM_F16X3=1 (split-fp16 MFMA, default) against =0 (fp32-input MFMA), HIP events, plus the
error of each against the fp64 oracle on a query sample."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402
from oracle import maxsim as OM  # noqa: E402

for lang in sys.argv[1:] or ["en"]:
    W = bench.build_corpus(lang, colbert=True)
    idx = _native.MaxSimIndex(W["D"], W["doc_ptr"], device=0)
    dev = torch.device("cuda", 0)
    Q = torch.from_numpy(W["Qtok"]).to(dev)
    nq, K = Q.shape[0], 10
    idx.reserve(nq, K)
    s = torch.empty((nq, K), dtype=torch.float32, device=dev)
    i = torch.empty((nq, K), dtype=torch.int64, device=dev)
    st = int(torch.cuda.current_stream().cuda_stream)
    tokens = int(W["doc_ptr"][-1])
    ref = OM.maxsim_scores(W["Qtok"][:48], W["D"], W["doc_ptr"])
    variants = [("1", "4", "8"), ("0", "4", "8"), ("1", "0", "8"), ("0", "0", "8"), ("1", "3", "8"), ("1", "6", "8"),
                ("1", "4", "16"), ("1", "4", "32"), ("1", "6", "32"), ("0", "4", "32"), ("1", "4", "8")]
    if os.environ.get("AB_VARIANTS"):
        variants = [tuple(v.split(":")) for v in os.environ["AB_VARIANTS"].split(",")]
    for var in variants:
        flag, ring, docs = var[:3]
        os.environ["AMDR_MAXSIM_TWOPASS"] = var[3] if len(var) > 3 else "1"
        os.environ["AMDR_MAXSIM_HI2"] = var[4] if len(var) > 4 else "1"
        os.environ["AMDR_MAXSIM_F16X3"] = flag
        os.environ["AMDR_MAXSIM_RING"] = ring
        os.environ["AMDR_MAXSIM_DOCS"] = docs
        ms = bench.event_ms(torch, lambda: idx.search_device(Q.data_ptr(), nq, 32, K, s.data_ptr(), i.data_ptr(), st), 5)
        got = idx.scores(W["Qtok"][:48])
        tf = 2.0 * 32 * 128 * tokens * nq / (ms * 1e-3) / 1e12
        print(f"{lang} F16X3={flag} ring={ring} docs/block={docs} twopass={os.environ['AMDR_MAXSIM_TWOPASS']} hi2={os.environ['AMDR_MAXSIM_HI2']}: {ms:.3f} ms per {nq} queries  {nq / ms * 1e3:,.0f} q/s  {tf:.1f} TFLOP/s-equivalent  "
              f"max|err| vs fp64 {np.max(np.abs(got - ref)):.2e}", flush=True)
    idx.close()
