"""The serving call as ONE launch (amdr_hybrid_small_device, csrc/fuse.hip hybrid_small_kernel): BM25 top-k + dense top-k +
fusion of 1-4 queries on a corpus of <= 2 048 chunks.  Checked against (a) the separate launches (AMDR_HYBRID_SMALL=0:
amdr_bm25_search_device + amdr_dense_search_fuse_device) — bit for bit, every output — and (b) the oracle directly
(oracle/dense.py, oracle/bm25.py, oracle/fuse.py: the reference's hybrid_retriever.py:181-209 + :389-551)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(rng, n, d, vocab):
    from oracle import bm25 as OB
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    words = [f"w{i}" for i in range(vocab)]
    docs = [[words[j] for j in rng.integers(0, vocab, size=int(rng.integers(3, 40)))] for _ in range(n)]
    ob = OB.BM25Okapi(docs)
    return X, ob, OB.to_csr(ob)


def _engine(X, ob, csr):
    from legal_rag_amd import _native
    from legal_rag_amd.retrieval.engine import HybridEngine
    return HybridEngine(_native.DenseIndex(X), _native.BM25Index(csr["term_ptr"], csr["post_doc"], csr["post_tf"], csr["idf"],
                                                                 csr["doc_len"], ob.avgdl, ob.k1, ob.b), None)


def _run(eng, params, k, Q, qt, qp, small):
    import torch
    old = os.environ.get("AMDR_HYBRID_SMALL")
    os.environ["AMDR_HYBRID_SMALL"] = "1" if small else "0"
    try:
        r = eng.search_batch(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)
        torch.cuda.synchronize()
        return {f: getattr(r, f).cpu().numpy().copy() for f in
                ("ids", "vals", "mask", "count", "dense_scores", "dense_ids", "bm25_scores", "bm25_ids")}
    finally:
        if old is None:
            os.environ.pop("AMDR_HYBRID_SMALL", None)
        else:
            os.environ["AMDR_HYBRID_SMALL"] = old


def _same(a, b, what):
    for f in a:
        x, y = a[f], b[f]
        if f == "count":
            assert np.array_equal(x, y), (what, f)
            continue
        if f in ("ids", "vals", "mask"):  # entries past count[q] are unspecified
            for q in range(x.shape[0]):
                c = int(a["count"][q])
                assert np.array_equal(x[q, :c].view(np.uint8), y[q, :c].view(np.uint8)), (what, f, q)
            continue
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (what, f)


@pytest.mark.parametrize("n,d", [(591, 384), (1260, 768), (2048, 64), (33, 384), (257, 1024), (1, 8)])
def test_one_launch_equals_the_separate_launches(n, d):
    import torch
    from legal_rag_amd import _native
    rng = np.random.default_rng(n * 7 + d)
    X, ob, csr = _corpus(rng, n, d, 150)
    eng = _engine(X, ob, csr)
    dev = torch.device("cuda", 0)
    V = len(csr["vocab"])
    for nq in (1, 2, 3, 4):
        for k in (1, 5, 10, 16):
            if k > n:
                continue
            q = rng.standard_normal((nq, d)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            toks = [[int(t) for t in rng.integers(-2, V, size=int(rng.integers(0, 24)))] for _ in range(nq)]
            qt_h, qp_h = _native.BM25Index.pack_queries(toks)
            Q = torch.from_numpy(q).to(dev)
            qt = torch.from_numpy(np.concatenate([qt_h, np.zeros(1, np.int32)])).to(dev)
            qp = torch.from_numpy(qp_h).to(dev)
            for method, mf in (("weighted_sum", 0.0), ("rrf", 0.0), ("weighted_sum", 0.2)):
                params = _native.make_fuse_params(method=method, min_final_score=mf)
                a = _run(eng, params, k, Q, qt, qp, True)
                a2 = _run(eng, params, k, Q, qt, qp, True)  # the arrival counters reset themselves
                b = _run(eng, params, k, Q, qt, qp, False)
                _same(a, b, (n, d, nq, k, method, mf))
                _same(a2, b, (n, d, nq, k, method, mf, "second launch"))


def test_one_launch_against_the_oracle_directly():
    import torch
    from legal_rag_amd import _native
    from oracle import dense as OD
    rng = np.random.default_rng(5)
    n, d, k = 1260, 768, 10
    X, ob, csr = _corpus(rng, n, d, 300)
    eng = _engine(X, ob, csr)
    dev = torch.device("cuda", 0)
    V = len(csr["vocab"])
    vocab = csr["vocab"]
    inv = {i: w for w, i in vocab.items()} if isinstance(vocab, dict) else {i: w for i, w in enumerate(vocab)}
    params = _native.make_fuse_params()
    for trial in range(6):
        nq = 1 + trial % 4
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        toks = [[int(t) for t in rng.integers(0, V, size=int(rng.integers(1, 20)))] for _ in range(nq)]
        qt_h, qp_h = _native.BM25Index.pack_queries(toks)
        a = _run(eng, params, k, torch.from_numpy(q).to(dev), torch.from_numpy(qt_h).to(dev), torch.from_numpy(qp_h).to(dev),
                 True)
        es, ei = OD.flatip_topk(X, q, k)
        assert np.array_equal(a["dense_ids"], ei)
        assert np.max(np.abs(a["dense_scores"] - es)) <= 1e-5
        for qi in range(nq):
            sc = np.asarray(ob.get_scores([inv[t] for t in toks[qi]]), dtype=np.float64)
            order = np.lexsort((np.arange(n), -sc))[:k]
            assert np.array_equal(a["bm25_ids"][qi], order), (trial, qi)
            assert np.array_equal(a["bm25_scores"][qi], sc[order]), (trial, qi)  # fp64, the reference's expression: bits


def test_one_launch_is_capturable_and_replays():
    import torch
    from legal_rag_amd import _native
    rng = np.random.default_rng(9)
    n, d, k = 591, 384, 10
    X, ob, csr = _corpus(rng, n, d, 200)
    eng = _engine(X, ob, csr)
    dev = torch.device("cuda", 0)
    V = len(csr["vocab"])
    params = _native.make_fuse_params(min_final_score=0.1)
    toks = [[int(t) for t in rng.integers(0, V, size=8)]]
    qt_h, qp_h = _native.BM25Index.pack_queries(toks)
    Q = torch.empty((1, d), dtype=torch.float32, device=dev)
    qt, qp = torch.from_numpy(qt_h).to(dev), torch.from_numpy(qp_h).to(dev)
    g, res = eng.capture(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)
    for seed in range(3):
        q = np.random.default_rng(seed).standard_normal((1, d)).astype(np.float32)
        Q.copy_(torch.from_numpy(q / np.linalg.norm(q)))
        qt.copy_(torch.from_numpy(np.random.default_rng(seed).integers(0, V, size=8).astype(np.int32)))
        g.replay()
        torch.cuda.synchronize()
        got = {f: getattr(res, f).cpu().numpy().copy() for f in
               ("ids", "vals", "mask", "count", "dense_scores", "dense_ids", "bm25_scores", "bm25_ids")}
        ref = _run(eng, params, k, Q, qt, qp, False)
        _same(got, ref, ("replay", seed))
