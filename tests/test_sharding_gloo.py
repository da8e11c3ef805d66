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


def test_shard_mode_without_a_process_group_fails_loudly():
    """cfg.retrieval.shard = "rows" outside torch.distributed is an error, never a silent whole-corpus index."""
    from legal_rag_amd.config import AppConfig
    cfg = AppConfig()
    assert sharding.active_shard(cfg.retrieval) is None  # default: unsharded
    cfg.retrieval.shard = "rows"
    with pytest.raises(RuntimeError, match="process group"):
        sharding.active_shard(cfg.retrieval)
    cfg.retrieval.shard = "columns"
    with pytest.raises(ValueError, match="only 'rows'"):
        sharding.active_shard(cfg.retrieval)


def test_shard_csr_partitions_the_postings():
    """The shards' posting lists are a partition of the whole index's: same (term, doc, tf) triples, local ids
    ascending per term, the term table of full length on every shard."""
    from legal_rag_amd.bm25_model import BM25Okapi, shard_csr
    rng = np.random.default_rng(3)
    docs = [[f"w{j}" for j in rng.integers(0, 50, size=int(rng.integers(1, 30)))] for _ in range(97)]
    tp, pd, pt, idf, dl = BM25Okapi(docs).to_csr()
    whole = sorted((t, int(pd[j]), int(pt[j])) for t in range(len(tp) - 1) for j in range(tp[t], tp[t + 1]))
    for world in (2, 3, 8):
        got = []
        for lo, hi in sharding.shard_bounds(len(docs), world):
            stp, spd, spt, sdl = shard_csr(tp, pd, pt, dl, lo, hi)
            assert len(stp) == len(tp) and stp[0] == 0 and stp[-1] == len(spd) == len(spt)
            assert sdl.tolist() == dl[lo:hi].tolist()
            for t in range(len(stp) - 1):
                seg = spd[stp[t]:stp[t + 1]]
                assert (np.diff(seg) > 0).all() and ((seg >= 0) & (seg < hi - lo)).all()
                got += [(t, int(d) + lo, int(f)) for d, f in zip(seg, spt[stp[t]:stp[t + 1]])]
        assert sorted(got) == whole


def _spec_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from legal_rag_amd.config import AppConfig
    cfg = AppConfig()
    cfg.retrieval.shard = "rows"
    spec = sharding.active_shard(cfg.retrieval)
    assert spec is not None and spec.rank == rank and spec.world == world and spec.key == (rank, world)
    assert spec.bounds(591) == sharding.shard_bounds(591, world)[rank]
    # three channels (f32, f64, f32) in ONE collective, as the sharded engine packs them
    rng = np.random.default_rng(11 + rank)
    nq, k = 5, 4
    chans = [(torch.from_numpy(np.sort(rng.standard_normal((nq, k)).astype(dt))[:, ::-1].copy()),
              torch.from_numpy(rng.integers(0, 50, size=(nq, k)))) for dt in (np.float32, np.float64, np.float32)]
    calls = []
    real = dist.all_gather_into_tensor

    def counting(*a, **kw):
        calls.append(1)
        return real(*a, **kw)
    dist.all_gather_into_tensor = counting
    out = sharding.exchange_topk(chans, 100 * rank, merge_fn=oracle_merge)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1 and len(out) == 3
    assert [o[0].dtype for o in out] == [torch.float32, torch.float64, torch.float32]
    torch.save([(s, i) for s, i in out], os.path.join(out_dir, f"x_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_active_shard_and_three_channel_exchange_world2_gloo(tmp_path):
    mp.spawn(_spec_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "x_0.pt"), torch.load(tmp_path / "x_1.pt")
    for (sa, ia), (sb, ib) in zip(a, b):
        assert torch.equal(sa, sb) and torch.equal(ia, ib)
        assert (ia >= 0).all() and (ia.max() >= 100)  # global ids: rank 1's carry its offset
        assert (sa[:, :-1] >= sa[:, 1:]).all()
