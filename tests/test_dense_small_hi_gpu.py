"""The fp16 first pass over a short corpus (csrc/dense_small_hi.hip, amdr_dense_small_*; experimental, not on the search
path): every approximate score lies within the PROVEN per-query bound of the exact dot product (fp64 here; the product's
fp32 scores are within 1e-6 of it), the bound is small enough to be useful, and a second pass that re-scores the rows
within 2 eps of the k-th best approximate score would see every row of the exact top-k (oracle/dense.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(X, Q):
    import torch
    from legal_rag_amd import _native
    dev = torch.device("cuda", 0)
    n, d = X.shape
    nq = Q.shape[0]
    idx = _native.DenseIndex(X, device=0)
    ap = _native.DenseSmallApprox(idx)
    ld = (n + 31) // 32 * 32
    S = torch.full((nq, ld), float("nan"), dtype=torch.float32, device=dev)
    eps = torch.empty((nq,), dtype=torch.float32, device=dev)
    Qd = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    ap.approx_device(Qd.data_ptr(), nq, S.data_ptr(), ld, eps.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    out = S.cpu().numpy(), eps.cpu().numpy()
    ap.close()
    idx.close()
    return out


@pytest.mark.parametrize("n,d,nq", [(591, 768, 300), (1260, 768, 70), (33, 128, 5), (1, 256, 1), (1024, 1024, 257), (600, 384, 64)])
def test_approximate_scores_stay_inside_the_proven_bound(n, d, nq):
    from oracle import dense as OD
    rng = np.random.default_rng(n * 3 + d + nq)
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    S, eps = _run(X, Q)
    exact = Q.astype(np.float64) @ X.astype(np.float64).T
    err = np.abs(S[:, :n] - exact)
    assert np.all(np.isfinite(S[:, :n])) and np.all(S[:, n:] == 0)
    assert np.all(err <= eps[:, None]), (float(err.max()), float(eps.min()))
    assert eps.max() < 3e-3 and err.max() < 2e-3  # unit vectors: ~1.1e-3 proven, ~2e-4 observed
    # the candidate rule of a second pass sees the exact top-k (ids of the fp32 oracle)
    k = min(10, n)
    _, ids = OD.flatip_topk(X, Q, k)
    for q in range(nq):
        tk = np.sort(S[q, :n])[::-1][k - 1]
        cand = set(np.nonzero(S[q, :n] >= tk - 2 * eps[q])[0].tolist())
        assert set(ids[q].tolist()) <= cand
        assert len(cand) <= max(64, 4 * k) or n <= 64


def test_scaled_operands_and_unusable_queries():
    rng = np.random.default_rng(7)
    n, d, nq = 200, 256, 40
    X = (rng.standard_normal((n, d)) * 37.5).astype(np.float32)
    Q = (rng.standard_normal((nq, d)) * np.float32(1e-4)).astype(np.float32)
    Q[3] = 0.0
    Q[5, 7] = np.nan
    Q[6, 0] = np.inf
    Q[7] *= np.float32(1e30)
    S, eps = _run(X, Q)
    exact = Q.astype(np.float64) @ X.astype(np.float64).T
    ok = np.ones(nq, bool)
    ok[[5, 6]] = False
    assert np.isnan(eps[5]) and np.isnan(eps[6]) and np.all(np.isfinite(eps[ok]))
    assert np.all(S[3, :n] == 0) and eps[3] >= 0
    err = np.abs(S[ok][:, :n] - exact[ok])
    assert np.all(err <= eps[ok][:, None])
    # relative to the scores' own scale the bound stays at the unit-vector level
    scale = np.linalg.norm(Q[ok].astype(np.float64), axis=1) * np.linalg.norm(X.astype(np.float64), axis=1).max()
    nz = scale > 0
    assert np.all(eps[ok][nz] <= 3e-3 * scale[nz] + 1e-30)
