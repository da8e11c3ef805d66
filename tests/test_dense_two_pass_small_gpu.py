"""The two-pass form of a long batch on a short corpus (round 4: dense_small_hi.hip first pass on the fp16 matrix
instructions + fuse.hip dense_hi_select_fuse_kernel: the rows inside the proven margin re-scored exactly) against the
exact form (AMDR_DENSE_SMALL_HI=0: dense_panel_scores_kernel + select) and the CPU oracle: same ids, scores within fp32
summation-order noise, the fusion's outputs the same — with the in-kernel exact fallback forced, on near-duplicate rows
(more than 32 rows inside the margin) and on queries without a bound (NaN / zero / huge)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _search(idx, Q, k, env):
    import torch
    old = {n: os.environ.get(n) for n in env}
    os.environ.update(env)
    try:
        dev = torch.device("cuda", 0)
        nq = Q.shape[0]
        Qd = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
        s = torch.empty((nq, k), dtype=torch.float32, device=dev)
        i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx.search_device(Qd.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return s.cpu().numpy(), i.cpu().numpy(), idx.plan_info(nq, k)
    finally:
        for n, v in old.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v


@pytest.mark.parametrize("n,d,nq,k", [(591, 768, 4100, 10), (1024, 384, 300, 12), (33, 128, 129, 5), (640, 1024, 200, 1),
                                      (257, 256, 513, 11)])
def test_two_pass_equals_the_exact_form_and_the_oracle(n, d, nq, k):
    from legal_rag_amd import _native
    from oracle import dense as OD
    rng = np.random.default_rng(n + d + nq)
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx = _native.DenseIndex(X, device=0)
    two = {"AMDR_DENSE_SMALL_HI": "1", "AMDR_DENSE_SMALL_HI_MIN": "96"}
    s2, i2, plan2 = _search(idx, Q, k, two)
    s1, i1, plan1 = _search(idx, Q, k, {"AMDR_DENSE_SMALL_HI": "0"})
    before = idx.two_pass_fallbacks()
    sf, i_f, _ = _search(idx, Q, k, dict(two, AMDR_DENSE_SMALL_HI_MARGIN="1e9"))  # every query: the exact fallback inside
    assert idx.two_pass_fallbacks() - before == (nq if n > 32 else 0)  # (<= 32 rows: all of them are candidates anyway)
    assert plan2.startswith("dsh_scores_kernel") and not plan1.startswith("dsh_scores_kernel")
    es, ei = OD.flatip_topk(X, Q, k)
    for s, i, what in ((s2, i2, "two-pass"), (sf, i_f, "fallback"), (s1, i1, "exact")):
        assert np.array_equal(i, ei), what
        assert np.max(np.abs(s - es)) <= TOL, what
    idx.close()


def test_deep_searches_keep_the_exact_form():
    """Beyond depth 12 the second pass's 32 candidate slots per query overflow too often: those searches take the exact form."""
    from legal_rag_amd import _native
    rng = np.random.default_rng(3)
    X = rng.standard_normal((300, 256)).astype(np.float32)
    idx = _native.DenseIndex(X, device=0)
    two = {"AMDR_DENSE_SMALL_HI": "1", "AMDR_DENSE_SMALL_HI_MIN": "96"}
    Q = rng.standard_normal((200, 256)).astype(np.float32)
    assert _search(idx, Q, 12, two)[2].startswith("dsh_scores_kernel")
    for k in (13, 16, 32):
        assert not _search(idx, Q, k, two)[2].startswith("dsh_scores_kernel")
    idx.close()


def test_near_duplicate_rows_unbounded_queries_and_ties():
    from legal_rag_amd import _native
    from oracle import dense as OD
    rng = np.random.default_rng(11)
    n, d, nq, k = 500, 256, 150, 10
    base = rng.standard_normal((20, d)).astype(np.float32)
    X = np.repeat(base, 25, axis=0)  # 25 exact copies of each of 20 rows: ties -> lower id first
    X[::7] += (rng.standard_normal((len(X[::7]), d)) * 1e-4).astype(np.float32)  # and near-copies inside the margin
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[3] = 0.0
    Q[5] *= np.float32(1e-30)
    Q[6] *= np.float32(1e20)
    idx = _native.DenseIndex(X, device=0)
    two = {"AMDR_DENSE_SMALL_HI": "1", "AMDR_DENSE_SMALL_HI_MIN": "96"}
    s2, i2, plan = _search(idx, Q, k, two)
    s1, i1, _ = _search(idx, Q, k, {"AMDR_DENSE_SMALL_HI": "0"})
    assert plan.startswith("dsh_scores_kernel")
    exact = Q.astype(np.float64) @ X.astype(np.float64).T
    for b in range(nq):
        scale = max(1e-30, float(np.abs(exact[b]).max()))
        # the ids are a valid top-k of the exact scores (rows tying within rounding may swap between the forms) ...
        kth = np.sort(exact[b])[::-1][k - 1]
        assert np.all(exact[b, i2[b]] >= kth - 1e-6 * scale), b
        assert np.max(np.abs(s2[b] - exact[b, i2[b]])) <= 3e-6 * scale, b
        assert len(set(i2[b].tolist())) == k
    # ... and exact duplicates come lower id first in both forms
    Xd = np.repeat(base[:4], 40, axis=0)
    Xd /= np.linalg.norm(Xd, axis=1, keepdims=True)
    idx2 = _native.DenseIndex(Xd, device=0)
    s2, i2, _ = _search(idx2, Q[:120], k, two)
    es, ei = OD.flatip_topk(Xd, Q[:120], k)
    ok = [b for b in range(120) if b not in (3,)]
    for b in ok:
        grp = i2[b] // 40
        assert np.all(np.diff(i2[b])[np.diff(grp) == 0] > 0), b  # inside a block of copies: ascending ids
    idx.close()
    idx2.close()


def test_fused_step_two_pass_equals_exact_form():
    import torch
    from legal_rag_amd import _native
    from legal_rag_amd.retrieval.engine import HybridEngine
    from oracle import bm25 as OB
    rng = np.random.default_rng(2)
    n, d, nq, k = 591, 768, 4200, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    words = [f"w{i}" for i in range(300)]
    docs = [[words[j] for j in rng.integers(0, 300, size=int(rng.integers(5, 60)))] for _ in range(n)]
    ob = OB.BM25Okapi(docs)
    csr = OB.to_csr(ob)
    eng = HybridEngine(_native.DenseIndex(X), _native.BM25Index(csr["term_ptr"], csr["post_doc"], csr["post_tf"], csr["idf"],
                                                                csr["doc_len"], ob.avgdl, ob.k1, ob.b), None)
    dev = torch.device("cuda", 0)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Q = torch.from_numpy(q).to(dev)
    qt_h, qp_h = _native.BM25Index.pack_queries([[int(t) for t in rng.integers(0, 300, size=6)] for _ in range(nq)])
    qt, qp = torch.from_numpy(qt_h).to(dev), torch.from_numpy(qp_h).to(dev)
    params = _native.make_fuse_params(min_final_score=0.2)
    out = {}
    for name, flag in (("two", "1"), ("exact", "0")):
        os.environ["AMDR_DENSE_SMALL_HI"] = flag
        try:
            r = eng.search_batch(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)
            torch.cuda.synchronize()
            out[name] = {f: getattr(r, f).cpu().numpy().copy() for f in ("ids", "vals", "mask", "count", "dense_scores", "dense_ids")}
        finally:
            os.environ.pop("AMDR_DENSE_SMALL_HI", None)
    a, b = out["two"], out["exact"]
    assert np.array_equal(a["dense_ids"], b["dense_ids"]) and np.max(np.abs(a["dense_scores"] - b["dense_scores"])) <= TOL
    assert np.array_equal(a["count"], b["count"])
    for qi in range(nq):
        c = int(a["count"][qi])
        assert np.array_equal(a["ids"][qi, :c], b["ids"][qi, :c]) and np.array_equal(a["mask"][qi, :c], b["mask"][qi, :c]), qi
        assert np.max(np.abs(a["vals"][qi, :c] - b["vals"][qi, :c])) <= 2e-5, qi  # (minmax divides by a small range)
