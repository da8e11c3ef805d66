"""Exact inner-product top-k — the function faiss.IndexFlatIP computes and the
reference's IndexHNSWFlat(METRIC_INNER_PRODUCT) approximates.

Follows: legalrag/retrieval/dense_retriever.py:41-44 (`index.search(q_vec, k)`
-> (scores f32[nq,k], ids i64[nq,k]) sorted by score descending, -1 padded when
k > ntotal, as faiss does) and faiss_builder.py:84-91 (metric = inner product
over L2-normalised rows).  Parity unpinned: faiss is absent here (see
oracle/__init__.py); tie order is fixed to "lower row id first".
"""
from __future__ import annotations

import numpy as np


def flatip_scores(X: np.ndarray, Q: np.ndarray) -> np.ndarray:
    """scores[b, i] = <Q[b], X[i]> in fp32 (BLAS sgemm)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    return Q @ X.T


def topk_desc(scores: np.ndarray, k: int):
    """Per-row top-k, score descending, ties -> lower index. -1/-inf padded."""
    nq, n = scores.shape
    k = int(k)
    out_s = np.full((nq, k), -np.inf, dtype=scores.dtype)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    kk = min(k, n)
    if kk == 0:
        return out_s, out_i
    for b in range(nq):
        row = scores[b]
        if kk < n:
            # candidates: everything >= the kk-th largest value (keeps all ties)
            kth = np.partition(row, n - kk)[n - kk]
            cand = np.nonzero(row >= kth)[0]
        else:
            cand = np.arange(n)
        order = np.lexsort((cand, -row[cand].astype(np.float64)))[:kk]
        sel = cand[order]
        out_s[b, :kk] = row[sel]
        out_i[b, :kk] = sel
    return out_s, out_i


def flatip_topk(X: np.ndarray, Q: np.ndarray, k: int):
    s, i = topk_desc(flatip_scores(X, Q), k)
    s = s.astype(np.float32)
    s[i < 0] = -np.finfo(np.float32).max  # faiss pads IP results with -FLT_MAX
    return s, i


def merge_topk(scores_parts, ids_parts, k: int):
    """Merge per-shard top-k lists (already carrying GLOBAL ids; -1 = padding).

    scores_parts/ids_parts: [R, nq, k_r] arrays.  Tie -> lower global id.
    """
    S = np.concatenate(list(scores_parts), axis=1)
    I = np.concatenate(list(ids_parts), axis=1).astype(np.int64)
    nq = S.shape[0]
    out_s = np.empty((nq, k), dtype=S.dtype)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    pad = -np.finfo(np.float32).max if S.dtype == np.float32 else -np.inf
    out_s[:] = pad
    for b in range(nq):
        valid = np.nonzero(I[b] >= 0)[0]
        order = np.lexsort((I[b, valid], -S[b, valid].astype(np.float64)))[:k]
        sel = valid[order]
        out_s[b, : len(sel)] = S[b, sel]
        out_i[b, : len(sel)] = I[b, sel]
    return out_s, out_i
