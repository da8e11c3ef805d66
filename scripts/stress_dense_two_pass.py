"""One-off stress of the two-pass long-batch dense form (dense_small_hi.hip + dense_hi_select_fuse_kernel): random corpus
shapes, batch sizes, depths, scales, blocks of duplicated / nearly duplicated rows; ids against the exact form
(AMDR_DENSE_SMALL_HI=0) and fp64 — a differing id must be a tie within fp32 noise.  python scripts/stress_dense_two_pass.py [cases]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import test_dense_two_pass_small_gpu as T  # noqa: E402
from legal_rag_amd import _native  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(2024)
    bad = swaps = 0
    for c in range(cases):
        n = int(rng.choice([1, 2, 31, 32, 33, 100, 255, 256, 257, 511, 591, 640, 641, 1000, 1024]))
        d = int(rng.choice([128, 256, 384, 512, 640, 768, 896, 1024]))
        nq = int(rng.integers(96, 700))
        k = int(rng.integers(1, min(12, n) + 1))  # (the two-pass form takes k <= 12)
        X = rng.standard_normal((n, d)).astype(np.float32)
        style = int(rng.integers(0, 4))
        if style == 1 and n >= 8:  # blocks of exact copies
            X = np.repeat(X[: (n + 7) // 8], 8, axis=0)[:n].copy()
        elif style == 2 and n >= 8:  # near copies inside the margin
            X = np.repeat(X[: (n + 7) // 8], 8, axis=0)[:n] + (rng.standard_normal((n, d)) * 1e-4).astype(np.float32)
        X = (X / np.linalg.norm(X, axis=1, keepdims=True) * np.float32(rng.choice([1.0, 1e-3, 50.0]))).astype(np.float32)
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True) * np.float32(rng.choice([1.0, 1e-4, 300.0]))).astype(np.float32)
        idx = _native.DenseIndex(X, device=0)
        s2, i2, plan = T._search(idx, Q, k, {"AMDR_DENSE_SMALL_HI": "1", "AMDR_DENSE_SMALL_HI_MIN": "96"})
        s1, i1, _ = T._search(idx, Q, k, {"AMDR_DENSE_SMALL_HI": "0"})
        idx.close()
        assert plan.startswith("dsh_scores_kernel"), plan
        if not np.array_equal(i1, i2):
            exact = Q.astype(np.float64) @ X.astype(np.float64).T
            for b in np.nonzero((i1 != i2).any(axis=1))[0]:
                scale = float(np.abs(exact[b]).max()) + 1e-300
                kth = np.sort(exact[b])[::-1][k - 1]
                ok = (np.all(exact[b, i2[b]] >= kth - 2e-6 * scale) and len(set(i2[b].tolist())) == k
                      and np.max(np.abs(s2[b] - exact[b, i2[b]])) <= 4e-6 * scale)
                swaps += 1
                if not ok:
                    bad += 1
                    print("MISMATCH case", c, (n, d, nq, k, style), "query", int(b), i1[b].tolist(), i2[b].tolist(), flush=True)
        if c % 50 == 0:
            print(f"case {c}: {bad} wrong, {swaps} tie swaps so far", flush=True)
    print(f"done: {cases} cases, {bad} wrong results, {swaps} queries whose ids differ by ties within fp32 noise")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
