"""N > 1 path on CPU: world_size-2 `gloo` processes exchange per-shard top-k
with the product's collective layer (retrieval/sharding.py).  The local search
and the merge are the ORACLE here (there is no GPU); the product default merge
is the HIP kernel and refuses CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from legal_rag_amd.retrieval import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_merge(scores, ids, k_out):
    from oracle import dense as OD
    s, i = OD.merge_topk(list(scores.numpy()), list(ids.numpy()), k_out)
    return torch.from_numpy(s), torch.from_numpy(i)


def _worker(rank, world, port, n, d, nq, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import bm25 as OB
    from oracle import dense as OD
    rng = np.random.default_rng(5)  # same data on every rank
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[n // 2] = X[3]  # exact cross-shard tie
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    docs = [[f"w{j}" for j in rng.integers(0, 40, size=int(rng.integers(1, 30)))] for _ in range(n)]
    qtok = [[f"w{j}" for j in rng.integers(0, 40, size=5)] for _ in range(nq)]
    lo, hi = sharding.shard_bounds(n, world)[rank]
    ds, di = OD.flatip_topk(X[lo:hi], Q, k)
    # BM25 with GLOBAL statistics, local documents
    ob = OB.BM25Okapi(docs)
    bs = np.full((nq, k), -np.inf)
    bi = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        full = ob.get_scores(qtok[q])[lo:hi]
        order = sorted(range(hi - lo), key=lambda i: full[i], reverse=True)[:k]
        bs[q, :len(order)] = full[order]
        bi[q, :len(order)] = order
    (gds, gdi), (gbs, gbi) = sharding.exchange_topk(
        [(torch.from_numpy(ds), torch.from_numpy(di)), (torch.from_numpy(bs), torch.from_numpy(bi))], lo,
        merge_fn=oracle_merge)
    es, ei = OD.flatip_topk(X, Q, k)
    assert np.array_equal(gdi.numpy(), ei), (rank, gdi.numpy()[0], ei[0])
    assert np.allclose(gds.numpy(), es, atol=1e-6)
    for q in range(nq):
        exp = OB.search(ob, qtok[q], k)
        assert gbi[q].tolist() == [e[0] for e in exp], (rank, q)
        assert gbs[q].tolist() == [e[1] for e in exp]
    np.save(os.path.join(out_dir, f"ids_{rank}.npy"), gdi.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_topk_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 101, 32, 6, 10, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "ids_0.npy")
    b = np.load(tmp_path / "ids_1.npy")
    assert np.array_equal(a, b)  # every rank ends with the identical global top-k


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 591, 10_000_000):
        for w in (1, 2, 3, 8):
            b = sharding.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_pack_unpack_roundtrip_is_bit_exact():
    rng = np.random.default_rng(0)
    s32 = torch.from_numpy(rng.standard_normal((4, 10)).astype(np.float32))
    s64 = torch.from_numpy(rng.standard_normal((4, 7)))
    i32 = torch.from_numpy(rng.integers(-1, 1000, size=(4, 10)))
    i64 = torch.from_numpy(rng.integers(-1, 1000, size=(4, 7)))
    buf = sharding.pack_channels([(s32, i32), (s64, i64)])
    assert buf.dtype == torch.int64 and buf.shape == (4, 34)
    (a, ai), (b, bi) = sharding.unpack_channels(buf.unsqueeze(0), [10, 7], [torch.float32, torch.float64])
    assert torch.equal(a[0], s32) and torch.equal(b[0], s64) and torch.equal(ai[0], i32) and torch.equal(bi[0], i64)
    assert torch.equal(sharding.to_global(torch.tensor([[-1, 0, 5]]), 100), torch.tensor([[-1, 100, 105]]))


def test_native_merge_refuses_cpu_tensors():
    with pytest.raises(RuntimeError):
        sharding.native_merge(torch.zeros((2, 1, 3)), torch.zeros((2, 1, 3), dtype=torch.int64), 3)
