#!/usr/bin/env python3
"""A/B of the batched dense scores kernels on one device, interleaved rounds in ONE process
(GPU box).  Variants are selected through the environment the library reads per launch:
    AMDR_DENSE_PANEL=0        32x32 wave tiles (dense_mfma.hip)
    AMDR_DENSE_PANEL=1        panel kernel (dense_panel.hip), planner's choice of parts
    AMDR_PANEL_PARTS=<p>      panel kernel with the row blocks cut into p parts
    AMDR_PANEL_PERSIST=0      one block per logical block instead of the persistent grid
Prints the median / min time of the scores kernel (handle profiling events) per variant and the
fp32 TFLOP/s it corresponds to, and checks every variant's top-k against torch fp32.

    python scripts/ab_dense_panel.py [rows] [queries] [dim] [rounds]
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from legal_rag_amd import _native  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 591
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 37376
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    k = 10
    _native.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    Q = torch.randn((nq, d), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    X = torch.randn((n, d), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    st = int(torch.cuda.current_stream().cuda_stream)
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
    s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    idx.reserve(nq, k)
    nb = (n + 15) // 16
    pmin = (nb + 7) // 8
    variants = [("tile32", {"AMDR_DENSE_PANEL": "0"}), ("panel", {"AMDR_DENSE_PANEL": "1"}),
                ("panel/1 per block", {"AMDR_DENSE_PANEL": "1", "AMDR_PANEL_PERSIST": "0"})]
    for qb in ("0",):
        for p in sorted(set([pmin, pmin + 1, pmin + 2, pmin + 3, pmin + 4, (nb + 4) // 5, (nb + 3) // 4, (nb + 2) // 3])):
            if pmin <= p <= nb:
                variants.append((f"panel/p{p}", {"AMDR_DENSE_PANEL": "1", "AMDR_PANEL_PARTS": str(p)}))
    ref = (Q[:128] @ X.T)
    top = torch.topk(ref, min(k, n), dim=1).indices.cpu().numpy()
    times = {name: [] for name, _ in variants}
    agree = {}
    for r in range(rounds + 1):
        for name, env in variants:
            for kk in ("AMDR_DENSE_PANEL", "AMDR_PANEL_PARTS", "AMDR_PANEL_PERSIST"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            steps = 5
            idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st)
            torch.cuda.synchronize()
            idx.profile_begin(steps)
            for _ in range(steps):
                idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st)
            torch.cuda.synchronize()
            ms, launches = idx.profile_end()
            if r > 0:
                times[name].append(ms / max(launches, 1))
            else:
                agree[name] = float(np.mean(top == i[:128, :top.shape[1]].cpu().numpy()))
    flops = 2.0 * n * d * nq
    print(f"rows {n} queries {nq} dim {d}: {flops / 1e9:.2f} GFLOP per launch")
    for name, _ in variants:
        t = sorted(times[name])
        med, mn = t[len(t) // 2], t[0]
        print(f"  {name:18s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {flops / (med * 1e-3) / 1e12:6.1f} TFLOP/s "
              f"({flops / (med * 1e-3) / 1e12 / 157.3:.3f} of fp32 MFMA peak)  id agreement {agree[name]:.4f}")
    idx.close()


if __name__ == "__main__":
    main()
