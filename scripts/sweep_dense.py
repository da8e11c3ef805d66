#!/usr/bin/env python3
"""Sweep of the dense scan kernel: queries-per-scan B on a synthetic N x 768 matrix
resident in HBM (development aid; bench.py is the contract benchmark)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", default="1,2,4,8,16,32")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    X = bench.synth_matrix(torch, a.rows, a.dim, dev)
    Q = torch.randn((1024, a.dim), device=dev)
    Q /= Q.norm(dim=1, keepdim=True)
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=a.rows, dim=a.dim, device=0, keepalive=X)
    st = int(torch.cuda.current_stream().cuda_stream)
    for B in [int(x) for x in a.batches.split(",")]:
        s = torch.empty((B, a.k), dtype=torch.float32, device=dev)
        i = torch.empty((B, a.k), dtype=torch.int64, device=dev)
        idx.reserve(B, a.k)
        for _ in range(2):
            idx.search_device(Q.data_ptr(), B, a.k, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        idx.profile_begin(a.steps)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            idx.search_device(Q.data_ptr(), B, a.k, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps
        ms, n = idx.profile_end()
        ms /= max(n, 1)
        gbs = a.rows * a.dim * 4 / (ms * 1e-3) / 1e9
        print(json.dumps({"B": B, "k": a.k, "scan_ms": round(ms, 4), "wall_ms": round(wall * 1e3, 4),
                          "GBs_per_pass_equiv": round(gbs, 1), "qps": round(B / wall, 1)}), flush=True)


if __name__ == "__main__":
    main()
