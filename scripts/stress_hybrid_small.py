"""One-off stress of the one-launch serving step (csrc/fuse.hip hybrid_small_kernel): random corpus shapes, depths, query
counts and token lists, every output compared bit for bit with the separate launches; repeated back-to-back launches on
one engine (the arrival counters reset themselves) and launches interleaved over two engines.  python scripts/stress_hybrid_small.py [cases]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import test_hybrid_small_gpu as T  # noqa: E402
from legal_rag_amd import _native  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(12345)
    dev = torch.device("cuda", 0)
    engines = []
    for _ in range(6):
        n = int(rng.choice([1, 2, 31, 64, 65, 300, 591, 1024, 1025, 1260, 2047, 2048]))
        d = int(rng.choice([8, 64, 128, 384, 768, 1024]))
        X, ob, csr = T._corpus(rng, n, d, int(rng.integers(5, 400)))
        engines.append((n, d, T._engine(X, ob, csr), len(csr["vocab"])))
    bad = 0
    for c in range(cases):
        n, d, eng, V = engines[int(rng.integers(0, len(engines)))]
        nq = int(rng.integers(1, 5))
        k = int(rng.integers(1, min(16, n) + 1))
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        toks = [[int(t) for t in rng.integers(-2, V, size=int(rng.integers(0, 40)))] for _ in range(nq)]
        qt_h, qp_h = _native.BM25Index.pack_queries(toks)
        Q = torch.from_numpy(q).to(dev)
        qt = torch.from_numpy(np.concatenate([qt_h, np.zeros(1, np.int32)])).to(dev)
        qp = torch.from_numpy(qp_h).to(dev)
        params = _native.make_fuse_params(method=str(rng.choice(["weighted_sum", "rrf", "rrf_norm_blend"])),
                                          min_final_score=float(rng.choice([0.0, 0.2, -1e9])))
        os.environ["AMDR_HYBRID_SMALL"] = "1"
        for _ in range(int(rng.integers(1, 6))):  # back-to-back, no synchronise in between
            eng.search_batch(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)
        a = T._run(eng, params, k, Q, qt, qp, True)
        b = T._run(eng, params, k, Q, qt, qp, False)
        try:
            T._same(a, b, (c, n, d, nq, k))
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
        if c % 100 == 0:
            print(f"case {c}: {bad} mismatches so far", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
