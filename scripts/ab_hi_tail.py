"""A/B of what follows the scan of a large dense search: the round-3 tail (AMDR_DENSE_HI_TAIL=0: flat candidate list,
~18 launches per 64-query pass) against the round-4 tail (per-query lists, 4 launches + 2 gated, one tail per <= 256
queries).  Default matrix = a 1/8 shard of BASELINE configs[4] (1.25 M x 768): what one rank of an 8-GPU node scans.
Per case: sum of the scan launches, wall per search, their difference (the tail), and whether ids and score bits are
those of the exact first pass (AMDR_DENSE_HI=0) on the full matrix."""
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


def run(X, Q, n, d, B, k, steps, env):
    old = {k_: os.environ.get(k_) for k_ in env}
    os.environ.update(env)
    try:
        idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
        idx.reserve(B, k)
        s = torch.empty((B, k), dtype=torch.float32, device="cuda")
        i = torch.empty((B, k), dtype=torch.int64, device="cuda")
        st = int(torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            idx.search_device(Q.data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        per = max(1, (B + 63) // 64)
        idx.profile_begin(steps * per + 8)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(steps):
            idx.search_device(Q.data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
        b.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        dev_ms = a.elapsed_time(b) / steps
        ms, launches = idx.profile_end()
        scan = ms / steps
        out = {"env": env, "B": B, "k": k, "plan": idx.plan_info(B, k)[:72], "scan_ms_per_search": round(scan, 4),
               "scan_launches_per_search": launches / steps, "wall_ms": round(wall, 4), "stream_ms": round(dev_ms, 4),
               "tail_us": round((dev_ms - scan) * 1e3, 1), "qps": round(B / (dev_ms * 1e-3)),
               "counters": list(idx.hi_counters())}
        res = (s.cpu().numpy().copy(), i.cpu().numpy().copy())
        idx.close()
        return out, res
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v


def main():
    n = int(os.environ.get("AB_N", 1_250_000))
    d = int(os.environ.get("AB_D", 768))
    steps = int(os.environ.get("AB_STEPS", 30))
    dev = torch.device("cuda:0")
    X = bench.synth_matrix(torch, n, d, dev, seed=1234)
    Q = bench.synth_queries(torch, dev, d)
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("AB_CASES", "64:10,256:10,32:10,64:40").split(",")]
    for B, k in cases:
        if os.environ.get("AB_ONLY"):  # one form only (whatever AMDR_DENSE_HI_TAIL the caller pinned): profiling runs
            o, _ = run(X, Q, n, d, B, k, steps, {})
            print(json.dumps(o), flush=True)
            continue
        ex, rx = run(X, Q, n, d, B, k, max(3, steps // 6), {"AMDR_DENSE_HI": "0"})
        o3, r3 = run(X, Q, n, d, B, k, steps, {"AMDR_DENSE_HI_TAIL": "0"})
        o4, r4 = run(X, Q, n, d, B, k, steps, {"AMDR_DENSE_HI_TAIL": "1"})
        same3 = bool(np.array_equal(rx[1], r3[1]) and np.array_equal(rx[0].view(np.uint32), r3[0].view(np.uint32)))
        same4 = bool(np.array_equal(rx[1], r4[1]) and np.array_equal(rx[0].view(np.uint32), r4[0].view(np.uint32)))
        print(json.dumps({"n": n, "d": d, "exact_first_pass": ex, "round3_tail": o3, "round4_tail": o4,
                          "round3_identical_to_exact": same3, "round4_identical_to_exact": same4}), flush=True)


if __name__ == "__main__":
    main()
