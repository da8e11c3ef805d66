"""Host-side BM25 index object: vocabulary, statistics and the CSR postings that
are uploaded to the GPU.  Stands where `rank_bm25.BM25Okapi` stands in the
reference (`BM25Retriever.bm25`, bm25_retriever.py:30,63), exposing the same
attribute names (k1, b, epsilon, corpus_size, avgdl, doc_freqs, idf, doc_len,
average_idf) so an index pickled by the reference loads into it and an index
built here unpickles under rank_bm25 (artifacts.py).

Statistics follow rank_bm25 0.2.2 exactly (see oracle/bm25.py for the cited
restatement): vocabulary in first-seen order, idf = log(N-df+0.5)-log(df+0.5)
with negative values floored to epsilon*mean(idf), avgdl = total tokens / N.
Scoring runs on the GPU (csrc/bm25.hip) — there is no host scoring path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native


def shard_csr(term_ptr, post_doc, post_tf, doc_len, lo: int, hi: int):
    """Doc-partitioned postings of the row block [lo, hi): every term keeps the postings of those documents only,
    renumbered to local ids 0 .. hi-lo-1 (ascending per term, as the whole index); the term table — and with it the
    caller's idf vector — keeps its full length, so term ids mean the same on every shard."""
    lo, hi = int(lo), int(hi)
    keep = (post_doc >= lo) & (post_doc < hi)
    cnt = np.zeros(len(term_ptr), dtype=np.int64)
    term_of = np.repeat(np.arange(len(term_ptr) - 1), np.diff(term_ptr))
    np.add.at(cnt, term_of[keep] + 1, 1)
    return (np.cumsum(cnt).astype(np.int64), (post_doc[keep] - lo).astype(np.int32), np.ascontiguousarray(post_tf[keep]),
            np.ascontiguousarray(doc_len[lo:hi]))


class BM25Okapi:
    def __init__(self, corpus: Optional[Sequence[Sequence[str]]] = None, tokenizer=None, k1: float = 1.5,
                 b: float = 0.75, epsilon: float = 0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.avgdl = 0.0
        self.doc_freqs: List[Dict[str, int]] = []
        self.idf: Dict[str, float] = {}
        self.doc_len: List[int] = []
        self.tokenizer = tokenizer
        self.average_idf = 0.0
        if corpus is not None:
            nd: Dict[str, int] = {}
            total = 0
            for document in corpus:
                self.doc_len.append(len(document))
                total += len(document)
                freqs: Dict[str, int] = {}
                for word in document:
                    freqs[word] = freqs.get(word, 0) + 1
                self.doc_freqs.append(freqs)
                for word in freqs:
                    nd[word] = nd.get(word, 0) + 1
                self.corpus_size += 1
            self.avgdl = total / self.corpus_size
            idf_sum = 0
            negative = []
            for word, freq in nd.items():
                v = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
                self.idf[word] = v
                idf_sum += v
                if v < 0:
                    negative.append(word)
            self.average_idf = idf_sum / len(self.idf)
            eps = self.epsilon * self.average_idf
            for word in negative:
                self.idf[word] = eps

    # -- device side --------------------------------------------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_vocab", "_gpu", "_gpu_device", "_tokenizer_id"):
            st.pop(k, None)
        return st

    def vocab(self) -> Dict[str, int]:
        v = self.__dict__.get("_vocab")
        if v is None:
            v = {w: t for t, w in enumerate(self.idf.keys())}
            self.__dict__["_vocab"] = v
        return v

    def to_csr(self):
        vocab = self.vocab()
        V = len(vocab)
        counts = np.zeros(V + 1, dtype=np.int64)
        for doc in self.doc_freqs:
            for w in doc:
                counts[vocab[w] + 1] += 1
        term_ptr = np.cumsum(counts).astype(np.int64)
        fill = term_ptr[:-1].copy()
        nnz = int(term_ptr[-1])
        post_doc = np.empty(nnz, dtype=np.int32)
        post_tf = np.empty(nnz, dtype=np.int32)
        for d, doc in enumerate(self.doc_freqs):
            for w, tf in doc.items():
                t = vocab[w]
                post_doc[fill[t]] = d
                post_tf[fill[t]] = tf
                fill[t] += 1
        idf = np.fromiter((self.idf[w] for w in vocab), dtype=np.float64, count=V)
        return term_ptr, post_doc, post_tf, idf, np.asarray(self.doc_len, dtype=np.int32)

    def gpu(self, device: int = 0, rows: Optional[tuple] = None) -> "_native.BM25Index":
        """The postings in HBM.  rows = (lo, hi): only the documents of that block (doc-partitioned postings at
        local ids 0 .. hi-lo-1) with the corpus-GLOBAL idf and avgdl — a row shard of a multi-GPU deployment
        (retrieval/sharding.py); per-document scores are then bit-identical to the unsharded index's."""
        key = (device, tuple(rows) if rows is not None else None)
        g = self.__dict__.get("_gpu")
        if g is None or self.__dict__.get("_gpu_device") != key:
            term_ptr, post_doc, post_tf, idf, doc_len = self.to_csr()
            if rows is not None:
                term_ptr, post_doc, post_tf, doc_len = shard_csr(term_ptr, post_doc, post_tf, doc_len, *rows)
            g = _native.BM25Index(term_ptr, post_doc, post_tf, idf, doc_len, float(self.avgdl), float(self.k1),
                                  float(self.b), device=device)
            self.__dict__["_gpu"] = g
            self.__dict__["_gpu_device"] = key
        return g

    def term_ids(self, tokens: Sequence[str]) -> List[int]:
        v = self.vocab()
        return [v.get(t, -1) for t in tokens]

    def get_scores(self, query: Sequence[str], device: int = 0) -> np.ndarray:
        """Same contract as rank_bm25's get_scores, computed by the HIP kernel."""
        return self.gpu(device).get_scores([self.term_ids(query)])[0]

    def top_k(self, query: Sequence[str], k: int, device: int = 0, shard=None):
        """(scores, doc ids) of the k best documents.  shard = sharding.ShardSpec: this rank scores its row block,
        the per-shard lists are all-gathered and merged; ids are global and identical on every rank."""
        if shard is None:
            s, i = self.gpu(device).search([self.term_ids(query)], k)
            return s[0], i[0]
        from .retrieval import sharding
        lo, hi = shard.bounds(self.corpus_size)
        s, i = self.gpu(device, rows=(lo, hi)).search([self.term_ids(query)], k)
        (gs, gi), = sharding.exchange_topk_numpy([(s, i)], lo, device, group=shard.group)
        return gs[0], gi[0]
