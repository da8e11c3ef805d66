"""ColBERT late-interaction MaxSim:  score(q, doc) = sum_i max_j <q_i, d_j>.

Follows: legalrag/retrieval/colbert_retriever.py:139-183 (pids/scores from
Searcher.search, pid == row of colbert_meta.jsonl) with the published ColBERT
scoring rule (colbert-ai `colbert_score`: per-query-token max over document
tokens, summed).  The build scores EXHAUSTIVELY over the shard on the token
embeddings it is given (PLAID's centroid pruning + 4-bit residual decompression
is replaced by design), so parity is vs this definition on
identical token embeddings.  Parity unpinned (colbert-ai absent here).
"""
from __future__ import annotations

import numpy as np

from .dense import topk_desc


def maxsim_scores(Q: np.ndarray, D: np.ndarray, doc_ptr: np.ndarray) -> np.ndarray:
    """Q: [nq, q_len, dim] ; D: [total_tokens, dim] ; doc_ptr: [n_docs+1].

    Returns fp64 scores [nq, n_docs] (computed in fp64 so the fp32 kernel is
    checked against a tighter reference).
    """
    Q = np.asarray(Q, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    nq = Q.shape[0]
    n_docs = len(doc_ptr) - 1
    out = np.empty((nq, n_docs), dtype=np.float64)
    for b in range(nq):
        S = Q[b] @ D.T  # [q_len, total_tokens]
        for d in range(n_docs):
            lo, hi = int(doc_ptr[d]), int(doc_ptr[d + 1])
            out[b, d] = S[:, lo:hi].max(axis=1).sum()
    return out


def maxsim_topk(Q, D, doc_ptr, k: int):
    s = maxsim_scores(Q, D, doc_ptr)
    return topk_desc(s, k)
