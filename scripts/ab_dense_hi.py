"""A/B of the large-scan first pass on the synthetic 10 M x 768 matrix: exact fp32 tile maxima (AMDR_DENSE_HI=0) against
the fp16 first pass (default).  Prints per batch size: scan-kernel ms, wall ms per search, queries/s, counters, and
whether ids and score bits of the two forms are identical on the full matrix."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from legal_rag_amd import _native  # noqa: E402


def run(X, Q, n, d, B, k, steps, hi):
    os.environ["AMDR_DENSE_HI"] = hi
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
    idx.reserve(B, k)
    s = torch.empty((B, k), dtype=torch.float32, device="cuda")
    i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        idx.search_device(Q.data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
    torch.cuda.synchronize()
    idx.profile_begin(steps * 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_device(Q.data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms, launches = idx.profile_end()
    out = {"hi": hi, "B": B, "k": k, "plan": idx.plan_info(B, k)[:60], "scan_ms": ms / max(launches, 1),
           "launches_per_search": launches / steps, "wall_ms": wall * 1e3, "qps": B / wall,
           "counters": list(idx.hi_counters())}
    res = (s.cpu().numpy().copy(), i.cpu().numpy().copy())
    idx.close()
    return out, res


def main():
    n = int(os.environ.get("AB_N", 10_000_000))
    d = int(os.environ.get("AB_D", 768))
    steps = int(os.environ.get("AB_STEPS", 5))
    dev = torch.device("cuda:0")
    X = bench.synth_matrix(torch, n, d, dev, seed=1234)
    Q = bench.synth_queries(torch, dev)
    if d != 768:
        Q = torch.nn.functional.normalize(torch.randn(1024, d, device=dev, generator=torch.Generator(dev).manual_seed(5)), dim=1)
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("AB_CASES", "4:10,32:10,64:10,128:10,64:40,5:10").split(",")]
    for B, k in cases:
        if os.environ.get("AB_ONLY_HI"):
            b, _ = run(X, Q, n, d, B, k, steps, "1")
            print(json.dumps(b), flush=True)
            continue
        a, ra = run(X, Q, n, d, B, k, steps, "0")
        b, rb = run(X, Q, n, d, B, k, steps, "1")
        same = bool(np.array_equal(ra[1], rb[1]) and np.array_equal(ra[0].view(np.uint32), rb[0].view(np.uint32)))
        print(json.dumps({"exact": a, "fp16_first_pass": b, "identical": same}), flush=True)


if __name__ == "__main__":
    main()
