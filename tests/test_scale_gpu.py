"""Size-independent properties at sizes the oracle cannot cover on the host
(BASELINE.json configs[4] shape, scaled to a few GB so the suite stays short):
shard-split + merge == unsharded, determinism, sortedness, and an oracle check on
a prefix read back from HBM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import torch
    from legal_rag_amd import _native
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    n, d = 2_000_000, 768
    X = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 500_000):
        blk = torch.randn((500_000, d), generator=g, device=dev)
        X[s:s + 500_000] = blk / blk.norm(dim=1, keepdim=True)
    Q = torch.randn((16, d), generator=g, device=dev)
    Q /= Q.norm(dim=1, keepdim=True)
    return torch, _native, X, Q


def search_dev(torch, _native, X, lo, hi, Q, k):
    idx = _native.DenseIndex(device_ptr=X[lo:hi].data_ptr(), n=hi - lo, dim=X.shape[1], device=0, keepalive=X)
    s = torch.empty((Q.shape[0], k), dtype=torch.float32, device=X.device)
    i = torch.empty((Q.shape[0], k), dtype=torch.int64, device=X.device)
    idx.search_device(Q.data_ptr(), Q.shape[0], k, s.data_ptr(), i.data_ptr(),
                      int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    idx.close()
    return s, i


@pytest.mark.parametrize("nq,k", [(1, 10), (8, 10), (16, 80)])
def test_sharded_merge_equals_unsharded(big, nq, k):
    torch, _native, X, Q = big
    from legal_rag_amd.retrieval import sharding
    n = X.shape[0]
    q = Q[:nq].contiguous()
    s_all, i_all = search_dev(torch, _native, X, 0, n, q, k)
    assert bool((s_all[:, :-1] >= s_all[:, 1:]).all())
    s2, i2 = search_dev(torch, _native, X, 0, n, q, k)
    assert torch.equal(s_all, s2) and torch.equal(i_all, i2)  # deterministic
    parts_s, parts_i = [], []
    for lo, hi in sharding.shard_bounds(n, 3):
        s, i = search_dev(torch, _native, X, lo, hi, q, k)
        parts_s.append(s)
        parts_i.append(sharding.to_global(i, lo))
    ms, mi = sharding.native_merge(torch.stack(parts_s), torch.stack(parts_i), k)
    torch.cuda.synchronize()
    assert torch.equal(mi, i_all) and torch.equal(ms, s_all)


def test_prefix_against_oracle(big):
    torch, _native, X, Q = big
    from oracle import dense as OD
    npre = 150_000
    s, i = search_dev(torch, _native, X, 0, npre, Q[:8].contiguous(), 10)
    es, ei = OD.flatip_topk(X[:npre].cpu().numpy(), Q[:8].cpu().numpy(), 10)
    assert np.array_equal(i.cpu().numpy(), ei)
    assert np.max(np.abs(s.cpu().numpy() - es)) <= 1e-4


def test_merge_f64_ties_prefer_lower_global_id(big):
    torch, _native, X, _ = big
    from legal_rag_amd.retrieval import sharding
    s = torch.tensor([[[3.0, 1.0, 0.0]], [[3.0, 2.0, 0.0]]], dtype=torch.float64, device=X.device)
    i = torch.tensor([[[40, 41, -1]], [[7, 8, 9]]], dtype=torch.int64, device=X.device)
    ms, mi = sharding.native_merge(s, i, 4)
    torch.cuda.synchronize()
    assert mi.tolist() == [[7, 40, 8, 41]] and ms.tolist() == [[3.0, 3.0, 2.0, 1.0]]
