"""N > 1 path with the REAL kernels: two `gloo` ranks on device 0 (RCCL refuses two ranks on one
card; on the driver's 8-GPU node the same code runs on backend nccl = RCCL).  Every rank searches
its row shard with the HIP kernels (dense + BM25 with global statistics), the per-shard top-k go
through sharding.exchange_topk (one all_gather_into_tensor of the packed lists), the merge is
merge_parts_kernel and the fusion fuse_kernel — and the result must equal the CPU oracle of the
UNSHARDED corpus on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from legal_rag_amd import _native
    from legal_rag_amd.retrieval import sharding
    from legal_rag_amd.retrieval.engine import HybridEngine
    from oracle import bm25 as OB
    from oracle import dense as OD
    from oracle import fusion as OF
    _native.load()
    rng = np.random.default_rng(9)  # same corpus and queries on every rank
    n, d, nq, K = 777, 128, 37, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[n // 2 + 3] = X[5]  # exact tie across the shard boundary: the lower global id must win everywhere
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    docs = [[f"w{j}" for j in rng.integers(0, 60, size=int(rng.integers(1, 40)))] for _ in range(n)]
    qtok = [[f"w{j}" for j in rng.integers(0, 60, size=6)] for _ in range(nq)]
    ob = OB.BM25Okapi(docs)
    csr = OB.to_csr(ob)
    lo, hi = sharding.shard_bounds(n, world)[rank]
    # this rank's shard: local rows, local postings, GLOBAL idf / avgdl
    tp, pd, pt = csr["term_ptr"], csr["post_doc"], csr["post_tf"]
    keep = (pd >= lo) & (pd < hi)
    cnt = np.zeros(len(tp), dtype=np.int64)
    np.add.at(cnt, np.repeat(np.arange(len(tp) - 1), np.diff(tp))[keep] + 1, 1)
    bmi = _native.BM25Index(np.cumsum(cnt), pd[keep] - lo, pt[keep], csr["idf"], csr["doc_len"][lo:hi], ob.avgdl,
                            ob.k1, ob.b, device=0)
    eng = HybridEngine(_native.DenseIndex(X[lo:hi], device=0), bmi, None, device=0)
    dev = torch.device("cuda", 0)
    tid = [[csr["vocab"].get(t, -1) for t in q] for q in qtok]
    qt, qp = _native.BM25Index.pack_queries(tid)
    q_emb, q_terms, q_ptr = (torch.from_numpy(a).to(dev) for a in (Q, qt, qp))
    d_loc = eng.dense_topk(q_emb, K)
    b_loc = eng.bm25_topk(q_terms, q_ptr, K)
    (gds, gdi), (gbs, gbi) = sharding.exchange_topk([d_loc, b_loc], lo)  # HIP merge on this rank
    params = _native.make_fuse_params(min_final_score=0.2)
    res = eng.fuse(params, nq, (gds, gdi), (gbs, gbi), None)
    torch.cuda.synchronize()
    # ---- the unsharded oracle ----
    es, ei = OD.flatip_topk(X, Q, K)
    assert np.array_equal(gdi.cpu().numpy(), ei), rank
    assert np.max(np.abs(gds.cpu().numpy() - es)) <= 1e-4
    ids, cnts = res.ids.cpu().numpy(), res.count.cpu().numpy()
    for q in range(nq):
        exp_b = OB.search(ob, qtok[q], K)
        assert gbi[q].tolist() == [e[0] for e in exp_b], (rank, q)   # BM25: bit-exact across the shard split
        assert gbs[q].tolist() == [e[1] for e in exp_b]
        dq = [(int(i), float(s)) for s, i in zip(gds[q].tolist(), gdi[q].tolist()) if i >= 0]
        fused = [h for h in OF.fuse(dq, exp_b, [], {}) if h["score"] >= 0.2]
        assert ids[q, :int(cnts[q])].tolist() == [h["id"] for h in fused], (rank, q)
    np.save(os.path.join(out_dir, f"fused_{rank}.npy"), ids)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_kernels_equal_unsharded_oracle(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "fused_0.npy"), np.load(tmp_path / "fused_1.npy")
    assert np.array_equal(a, b)  # every rank ends with the identical fused lists
