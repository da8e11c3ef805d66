#!/usr/bin/env python3
"""Dense batched search on mid-size corpora (GPU box): rows in {591 .. 30 000}, 8 192 queries,
d = 768.  Prints the wall time of the whole search (scores + top-k [+ merge]) and the share of
the scores kernel (handle profiling events), and checks a sample against numpy.

    python scripts/sweep_dense_mid.py [k]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from legal_rag_amd import _native  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    _native.load()
    dev = torch.device("cuda", 0)
    nq, d = 8192, 768
    g = torch.Generator(device=dev).manual_seed(1)
    Q = torch.randn((nq, d), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    st = int(torch.cuda.current_stream().cuda_stream)
    for n in (591, 1260, 2048, 3000, 9000, 30000):
        X = torch.randn((n, d), device=dev, generator=g)
        X /= X.norm(dim=1, keepdim=True)
        idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
        s = torch.empty((nq, k), dtype=torch.float32, device=dev)
        i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx.reserve(nq, k)
        for _ in range(3):
            idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        steps = 20
        idx.profile_begin(steps)
        t = time.perf_counter()
        for _ in range(steps):
            idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / steps
        ms, launches = idx.profile_end()
        ref = (Q[:64] @ X.T).cpu().numpy()
        top = np.argsort(-ref, axis=1, kind="stable")[:, :k]
        agree = float(np.mean(top == i[:64].cpu().numpy()))
        print(f"rows {n:6d} k {k}: search {wall * 1e6:8.1f} us, scores kernel {ms / max(launches, 1) * 1e3:8.1f} us, "
              f"top-k (+merge) {wall * 1e6 - ms / max(launches, 1) * 1e3:8.1f} us, id agreement {agree:.4f}")
        idx.close()


if __name__ == "__main__":
    main()
