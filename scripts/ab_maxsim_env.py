#!/usr/bin/env python3
"""Interleaved same-process A/B of the MaxSim channel under environment pins (the launch code reads them per call):

    python scripts/ab_maxsim_env.py en "AMDR_MAXSIM_DOCS=32" "AMDR_MAXSIM_DOCS=29" ""      ("" = defaults)

Every variant is timed (HIP events, 10 searches) once per round, in turn, for AB_ROUNDS rounds — box-to-box and
minute-to-minute drift (several per cent on a power-limited part) hits all variants alike; reported: median and min."""
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402


def main():
    lang = sys.argv[1]
    variants = sys.argv[2:] or [""]
    W = bench.build_corpus(lang, colbert=True)
    idx = _native.MaxSimIndex(W["D"], W["doc_ptr"], device=0)
    dev = torch.device("cuda", 0)
    Q = torch.from_numpy(W["Qtok"]).to(dev)
    nq, K = Q.shape[0], 10
    idx.reserve(nq, K)
    s = torch.empty((nq, K), dtype=torch.float32, device=dev)
    i = torch.empty((nq, K), dtype=torch.int64, device=dev)
    st = int(torch.cuda.current_stream().cuda_stream)
    run = lambda: idx.search_device(Q.data_ptr(), nq, Q.shape[1], K, s.data_ptr(), i.data_ptr(), st)  # noqa: E731
    times = {v: [] for v in variants}
    ref = None
    for rnd in range(int(os.environ.get("AB_ROUNDS", 7))):
        for v in variants:
            pins = dict(kv.split("=", 1) for kv in v.split(",") if kv)
            old = {k: os.environ.get(k) for k in pins}
            os.environ.update(pins)
            try:
                run()
                torch.cuda.synchronize()
                times[v].append(bench.event_ms(torch, run, 10))
                got = (s.cpu().numpy().copy(), i.cpu().numpy().copy())
                if ref is None:
                    ref = got
                assert (got[1] == ref[1]).all() and (got[0].view("uint32") == ref[0].view("uint32")).all(), v
            finally:
                for k, o in old.items():
                    if o is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = o
    for v in variants:
        t = times[v]
        print(f"{lang} {v or '(defaults)':60s} median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}", flush=True)


if __name__ == "__main__":
    main()
