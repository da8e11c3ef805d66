#!/usr/bin/env python3
"""Time of the experimental fp16 first pass on the headline step's shape (37 376 queries x 591 chunks x 768) beside the
exact scores kernel of the same step and library GEMMs (torch.matmul) of the same shape; HIP events, one process."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 591, int(sys.argv[2]) if len(sys.argv) > 2 else 768
out = {"n": n, "d": d}
X = torch.randn((n, d), device=dev, generator=g)
X /= X.norm(dim=1, keepdim=True)
idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
ap = _native.DenseSmallApprox(idx)
ld = (n + 31) // 32 * 32
st = int(torch.cuda.current_stream().cuda_stream)
import os
for nq in (1168, 2048, 4096, 8192, 9344, 37376):
    Q = torch.randn((nq, d), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    S = torch.empty((nq, ld), device=dev)
    eps = torch.empty((nq,), device=dev)
    t_ap = bench.event_ms(torch, lambda: ap.approx_device(Q.data_ptr(), nq, S.data_ptr(), ld, eps.data_ptr(), st), 30) * 1e3
    k = 10
    s = torch.empty((nq, k), device=dev)
    i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    idx.reserve(nq, k)
    os.environ["AMDR_DENSE_SMALL_HI"] = "0"
    t_exact = bench.event_ms(torch, lambda: idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st), 30) * 1e3
    os.environ["AMDR_DENSE_SMALL_HI"] = "1"
    os.environ["AMDR_DENSE_SMALL_HI_MIN"] = "96"
    t_two = bench.event_ms(torch, lambda: idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st), 30) * 1e3
    os.environ.pop("AMDR_DENSE_SMALL_HI_MIN")
    Qh, Xh = Q.half(), X.half()
    Sh = torch.empty((nq, n), device=dev, dtype=torch.half)
    t_lib = bench.event_ms(torch, lambda: torch.matmul(Qh, Xh.t(), out=Sh), 30) * 1e3
    t_cvt = bench.event_ms(torch, lambda: Qh.copy_(Q), 30) * 1e3
    err = (S[:, :n].double() - Q.double() @ X.double().t()).abs()
    out[str(nq)] = {"approx_pass_us": round(t_ap, 1), "exact_search_top10_us (scores + top-k)": round(t_exact, 1), "two_pass_search_top10_us": round(t_two, 1),
                    "library_fp16_gemm_us": round(t_lib, 1), "library_fp32_to_fp16_of_Q_us": round(t_cvt, 1),
                    "max_abs_err": float(err.max()), "eps_min_max": [float(eps.min()), float(eps.max())],
                    "inside_bound": bool((err <= eps[:, None]).all())}
print(json.dumps(out))
