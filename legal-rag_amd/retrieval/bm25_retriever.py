"""Sparse channel (legalrag/retrieval/bm25_retriever.py:19-76).

`load()` unpickles {"bm25", "chunks"} (RuntimeError when the file is missing or
has no "bm25", mtime-guarded reload); `search()` tokenises the query with
jieba.cut WITHOUT lower-casing (:73), scores every document with Okapi BM25 and
returns the first top_k of a stable descending sort — zero-score documents
included, ties in ascending document order (:74-76).  Scoring and ranking run in
the HIP kernel (csrc/bm25.hip), bit-identical to rank_bm25's float64 numpy
expression.

Tokenisation is never silently inexact (text.py): a Han query without a
segmenter raises `text.ZhTokenizerUnavailable` unless cfg.retrieval.zh_tokenizer
= "char" opts in (then `zh_exact` is False and the hybrid layer stamps it into
score_breakdown); `search(..., tokens=)` takes the caller's own tokens; an index
whose recorded tokenizer is "char" is queried in the same mode so that index and
query tokens stay consistent."""
from __future__ import annotations

import threading
from pathlib import Path
import logging
from typing import List, Optional, Sequence, Tuple

from .. import artifacts, text
from ..bm25_model import BM25Okapi
from ..schemas import LawChunk


logger = logging.getLogger(__name__)


class BM25Retriever:
    def __init__(self, cfg):
        self.cfg = cfg
        rcfg = cfg.retrieval
        self.bm25_path = Path(rcfg.bm25_index_file)
        self.device_index = int(getattr(rcfg, "device", 0))
        self._loaded = False
        self._bm25_mtime: float | None = None
        self.bm25: BM25Okapi | None = None
        self.chunks: List[LawChunk] = []
        self._lock = threading.Lock()
        self.index_tokenizer: Optional[str] = None  # id recorded in bm25.pkl (None: reference-built)
        self._tls = threading.local()                 # per-thread: did the last query use the stand-in?
        self.shard = None                             # sharding.ShardSpec in a row-sharded deployment

    def load(self) -> None:
        if not self.bm25_path.exists():
            raise RuntimeError(
                f"[BM25] index not found: {self.bm25_path}. "
                f"Run: python build_index.py (or python build_index.py --only-bm25)")
        current_mtime = self.bm25_path.stat().st_mtime
        if self._loaded and self._bm25_mtime == current_mtime:
            return
        with self._lock:
            if self._loaded and self._bm25_mtime == current_mtime:
                return
            bm25, chunks = artifacts.read_bm25_pickle(self.bm25_path)
            from . import sharding
            self.shard = sharding.active_shard(self.cfg.retrieval)  # row-sharded deployment: this rank's block only
            # upload CSR postings now, not on the first query
            bm25.gpu(self.device_index, rows=self.shard.bounds(bm25.corpus_size) if self.shard else None)
            self.bm25 = bm25
            self.chunks = chunks
            self.index_tokenizer = bm25.__dict__.get("_tokenizer_id")
            if self.index_tokenizer == "char":
                logger.warning("[BM25] %s was built with the one-character stand-in tokenizer: results differ "
                               "from a jieba-built index (zh_exact=False)", self.bm25_path)
            self._loaded = True
            self._bm25_mtime = current_mtime

    def gpu_index(self):
        """The _native.BM25Index this retriever scores with (this rank's row block in a sharded deployment)."""
        self.load()
        return self.bm25.gpu(self.device_index, rows=self.shard.bounds(self.bm25.corpus_size) if self.shard else None)

    def tokenize_query(self, query: str) -> List[str]:
        """bm25_retriever.py:73 — jieba.cut, not lower-cased.  Mode "char" when the index was built
        that way or the config opts in; otherwise a Han query without a segmenter raises."""
        mode = "char" if self.index_tokenizer == "char" else text.cfg_mode(self.cfg)
        if self.index_tokenizer == "char" and text.contains_han(query):
            toks = text.jieba_cut_restated(query)  # same stand-in as the index, whatever is installed
        else:
            toks = text.jieba_cut(query, mode)
        self._tls.exact = not (text.contains_han(query) and (self.index_tokenizer == "char" or not text.zh_exact()))
        return toks

    def term_ids_batch(self, questions: Sequence[str]):
        """Tokenise + look up a whole batch: (q_terms i32, q_ptr i64 [n+1], exact bool [n]) — the query CSR of
        amdr_bm25_search.  Text without Han characters goes through the native batched tokeniser (one call, GIL
        released: amdr_tokenizer_encode, token for token what text.jieba_cut returns for it); a query holding Han
        characters — and every query when jieba or a registered segmenter is present — takes tokenize_query()."""
        import numpy as np
        from .. import _native
        self.load()
        n = len(questions)
        # A registered segmenter sees every query.  With jieba installed — every real deployment of the reference — text
        # without Han characters still takes the native tokeniser (jieba decides it without its dictionary, text.py),
        # EXCEPT a query that holds one of the ASCII entries of jieba's dictionary (text._ASCII_DICT_WORDS: the one place
        # where the dictionary reaches into ASCII text): that query, like a Han one, goes to jieba itself.  The fallback is
        # per query, not per process (round 3 sent every English query through per-query Python once jieba was importable).
        native_ok = text._custom_cut is None
        if native_ok:
            tok = self.__dict__.get("_native_tok")
            if tok is None or tok[0] is not self.bm25:
                tok = (self.bm25, _native.Tokenizer(list(self.bm25.vocab().keys())))
                self.__dict__["_native_tok"] = tok
            qs = questions  # (None counts as the empty query inside the native call)
            terms, q_ptr, hard = tok[1].encode(qs)
            if text.HAVE_JIEBA:
                qs = [q or "" for q in questions]
                joined = "\0".join(qs)
                if any(w in joined for w in text._ASCII_DICT_WORDS):  # rare: find the queries concerned
                    hard = hard | np.fromiter((any(w in q for w in text._ASCII_DICT_WORDS) for q in qs), dtype=bool, count=n)
            # text without Han characters is tokenised exactly whatever the index was built with (tokenize_query's rule:
            # only a Han query on a char-built index, or without jieba, is a stand-in result)
            exact = np.ones(n, dtype=bool)
            if not hard.any():
                return terms, q_ptr, exact
        else:
            hard = np.ones(n, dtype=bool)
            terms, q_ptr = np.zeros(0, np.int32), np.zeros(n + 1, np.int64)
            exact = np.ones(n, dtype=bool)
        # the remaining queries one by one, spliced into the CSR
        parts, lens = [], np.diff(q_ptr)
        for i in range(n):
            if hard[i]:
                ids = np.asarray(self.bm25.term_ids(self.tokenize_query(questions[i])), dtype=np.int32)
                exact[i] = self.zh_exact
                lens[i] = len(ids)
                parts.append(ids)
            else:
                parts.append(terms[q_ptr[i]:q_ptr[i + 1]])
        out_ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=out_ptr[1:])
        return (np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)), out_ptr, exact

    @property
    def zh_exact(self) -> bool:
        """False when this thread's last query (or the index itself) went through the
        one-character stand-in instead of jieba."""
        return self.index_tokenizer != "char" and getattr(self._tls, "exact", True)

    def search(self, query: str, top_k: int, tokens: Optional[Sequence[str]] = None) -> List[Tuple[LawChunk, float]]:
        self.load()
        assert self.bm25 is not None
        if tokens is not None:
            tokens = list(tokens)
            self._tls.exact = True
        else:
            tokens = self.tokenize_query(query)
        k = int(top_k)
        if k <= 0:
            return []
        out: List[Tuple[LawChunk, float]] = []
        n = len(self.chunks)
        # the kernel ranks at most MAX_K per call; deeper requests take the dense score vector
        from .._native import MAX_K
        if k <= MAX_K:
            scores, ids = self.bm25.top_k(tokens, min(k, MAX_K), device=self.device_index, shard=self.shard)
            for s, i in zip(scores.tolist(), ids.tolist()):
                if i < 0:
                    break
                out.append((self.chunks[i], float(s)))
            return out
        if self.shard is not None:
            raise ValueError(f"sharded BM25 search: depth {k} exceeds the kernels' limit of {MAX_K}")
        scores = self.bm25.get_scores(tokens, device=self.device_index)
        idxs = sorted(range(n), key=lambda i: scores[i], reverse=True)[:k]
        return [(self.chunks[i], float(scores[i])) for i in idxs]
