"""BM25Okapi restatement (rank_bm25 0.2.2, pure Python/numpy float64).

Follows the reference call sites:
  legalrag/retrieval/builders/bm25_builder.py:18-19,39-44  (index tokenisation,
      `BM25Okapi(corpus_tokens)` with the wheel's defaults k1=1.5 b=0.75
      epsilon=0.25)
  legalrag/retrieval/bm25_retriever.py:73-76  (`get_scores(tokens)`, then
      `sorted(range(N), key=scores[i], reverse=True)[:top_k]` — a STABLE sort,
      so ties keep ascending doc index and zero-score docs ARE returned)

rank_bm25 is a requirements.txt dependency (`rank-bm25>=0.2.2`, unpinned, not
vendored, absent from the build container); its published algorithm is
restated below operation by operation so that float64 results are bit-equal:
  idf(t)   = log(N - df + 0.5) - log(df + 0.5)            (math.log, per word,
             in first-seen word order; idf_sum accumulated in that order)
  negative idf -> epsilon * (idf_sum / |vocab|)
  avgdl    = sum(len(doc)) / N
  score[d] += idf(t) * ( tf*(k1+1) / ( tf + k1*(1 - b + b*len[d]/avgdl) ) )
             once per query token, in query order, duplicates counted,
             unknown tokens contribute idf 0.
Pinned only weakly: the wheel is absent and the reference's tests hold no BM25
numbers, so the one external known answer is the example printed in rank_bm25's own
README ("windy London" over three sentences -> [0, 0.93729472, 0], 8 digits; it
exercises the idf formula, the epsilon floor and the length normalisation) —
tests/test_oracle_selfcheck.py::test_bm25_matches_rank_bm25_readme_example.  Beyond
that digit count parity with the wheel is unpinned (see oracle/__init__.py); a
hand-checkable toy vector lives in tests/golden/bm25_toy.json.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Sequence

import numpy as np

_EN_TOKEN_RE = re.compile(r"[A-Za-z0-9]+(?:'[A-Za-z0-9]+)?")


def tokenize_en(text: str) -> List[str]:
    """Index-side English tokeniser (bm25_builder.py:18-19)."""
    return _EN_TOKEN_RE.findall(text.lower())


class BM25Okapi:
    def __init__(self, corpus: Sequence[Sequence[str]], k1: float = 1.5, b: float = 0.75,
                 epsilon: float = 0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.avgdl = 0.0
        self.doc_freqs: List[Dict[str, int]] = []
        self.idf: Dict[str, float] = {}
        self.doc_len: List[int] = []
        nd = self._initialize(corpus)
        self._calc_idf(nd)

    def _initialize(self, corpus):
        nd: Dict[str, int] = {}
        num_doc = 0
        for document in corpus:
            self.doc_len.append(len(document))
            num_doc += len(document)
            frequencies: Dict[str, int] = {}
            for word in document:
                if word not in frequencies:
                    frequencies[word] = 0
                frequencies[word] += 1
            self.doc_freqs.append(frequencies)
            for word in frequencies:
                nd[word] = nd.get(word, 0) + 1
            self.corpus_size += 1
        self.avgdl = num_doc / self.corpus_size
        return nd

    def _calc_idf(self, nd):
        idf_sum = 0
        negative_idfs = []
        for word, freq in nd.items():
            idf = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
            self.idf[word] = idf
            idf_sum += idf
            if idf < 0:
                negative_idfs.append(word)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for word in negative_idfs:
            self.idf[word] = eps

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(doc.get(q) or 0) for doc in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (
                q_freq * (self.k1 + 1) / (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl))
            )
        return score


def search(bm25: BM25Okapi, tokens: Sequence[str], top_k: int):
    """bm25_retriever.py:74-76: full stable descending sort, first top_k."""
    scores = bm25.get_scores(tokens)
    idxs = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)[: int(top_k)]
    return [(int(i), float(scores[i])) for i in idxs]


# ---------------------------------------------------------------------------
# Sparse (CSR) view used to hand the same index to the HIP kernel in tests.
# ---------------------------------------------------------------------------
def to_csr(bm25: BM25Okapi):
    """Term-major CSR: vocab in first-seen order (== bm25.idf key order).

    Returns dict(vocab, term_ptr i64[V+1], post_doc i32[nnz], post_tf i32[nnz],
    idf f64[V], doc_len i32[N]).  Postings of a term are in ascending doc id.
    """
    vocab = {w: t for t, w in enumerate(bm25.idf.keys())}
    V = len(vocab)
    counts = np.zeros(V + 1, dtype=np.int64)
    for doc in bm25.doc_freqs:
        for w in doc:
            counts[vocab[w] + 1] += 1
    term_ptr = np.cumsum(counts)
    fill = term_ptr[:-1].copy()
    nnz = int(term_ptr[-1])
    post_doc = np.empty(nnz, dtype=np.int32)
    post_tf = np.empty(nnz, dtype=np.int32)
    for d, doc in enumerate(bm25.doc_freqs):
        for w, tf in doc.items():
            t = vocab[w]
            post_doc[fill[t]] = d
            post_tf[fill[t]] = tf
            fill[t] += 1
    idf = np.array([bm25.idf[w] for w in vocab], dtype=np.float64)
    return dict(vocab=vocab, term_ptr=term_ptr.astype(np.int64), post_doc=post_doc, post_tf=post_tf,
                idf=idf, doc_len=np.array(bm25.doc_len, dtype=np.int32))
