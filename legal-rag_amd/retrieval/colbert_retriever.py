"""ColBERT channel (counterpart of legalrag/retrieval/colbert_retriever.py:30-183).

Contract kept: `ColBERTRetriever(cfg)` / `.from_config(cfg)` (one instance per index
identity), `.enabled`, `search(query, top_k=5) -> [(LawChunk, score)]`;
RuntimeError when colbert_meta.jsonl is missing (HybridRetriever swallows it at
construction), `[]` when disabled, for a blank query, or on GPU out-of-memory.

The reference delegates to colbert-ai's PLAID Searcher (centroid pruning + 4-bit
residuals).  Here the query's [32, 128] token matrix is scored EXHAUSTIVELY
against the fp32 token embeddings of every document with the MaxSim HIP kernel —
a Legal-RAG corpus shard fits HBM thousands of times over, so nothing is pruned
and nothing is quantised.  pid == row of colbert_meta.jsonl, as the reference's
builder writes it (colbert_builder.py:39-52).
"""
from __future__ import annotations

import os
import threading
from pathlib import Path
from typing import ClassVar, Dict, List, Optional, Tuple

import numpy as np

from .. import _native, artifacts, encoders
from ..schemas import LawChunk

_TOKEN_ENCODERS: Dict[Tuple[str, str, int], object] = {}


def get_token_encoder(model_name: Optional[str], backend: str, doc_maxlen: int, device: Optional[str] = None):
    """'hashing' -> deterministic stand-in; a local checkpoint directory -> BERT-style
    ColBERT on PyTorch-ROCm; anything else cannot be had offline and fails loudly."""
    key = (str(model_name), backend, int(doc_maxlen), device)
    enc = _TOKEN_ENCODERS.get(key)
    if enc is None:
        if backend == "hashing":
            enc = encoders.HashingTokenEmbedder(doc_maxlen=doc_maxlen)
        elif model_name and os.path.isdir(str(model_name)):
            enc = encoders.TransformersColBERT(str(model_name), doc_maxlen=doc_maxlen, device=device)
        else:
            raise RuntimeError(
                f"ColBERT checkpoint '{model_name}' cannot be loaded offline (jina-colbert-v2 needs remote code and "
                f"a download); set cfg.retrieval.encoder_backend='hashing' for the deterministic stand-in encoder.")
        _TOKEN_ENCODERS[key] = enc
    return enc


def _identity(rcfg) -> Tuple[str, str, str, str, str, int]:
    g = lambda name, default=None: getattr(rcfg, name, default)  # noqa: E731
    return (str(g("colbert_index_path")), str(g("colbert_index_name")),
            str(g("colbert_model_name", "colbert-ir/colbertv2.0")), str(g("colbert_meta_file")),
            str(g("colbert_experiment")), int(g("colbert_nranks", 1)))


class ColBERTRetriever:
    _instances_by_key: ClassVar[Dict[tuple, "ColBERTRetriever"]] = {}
    _searcher_cache: ClassVar[Dict[tuple, Tuple["_native.MaxSimIndex", int]]] = {}  # (index, first pid of the shard)
    # from_config() constructs an instance (which fills the searcher cache) while holding the
    # registry lock: re-entrant on purpose
    _registry_lock: ClassVar[threading.RLock] = threading.RLock()

    def __init__(self, cfg):
        rcfg = cfg.retrieval
        self.cfg = cfg
        self.enabled = bool(getattr(rcfg, "enable_colbert", False))
        (index_path, self.index_name, self.model_name, meta_file, self.experiment, self.nranks) = _identity(rcfg)
        self.index_path, self.meta_file = Path(index_path), Path(meta_file)
        self.device_index = int(getattr(rcfg, "device", 0))
        self._pid2chunk: Dict[int, LawChunk] = {}
        self._collection: List[str] = []
        self._meta_mtime: Optional[float] = None
        self._searcher: Optional[_native.MaxSimIndex] = None
        self.shard, self.row_offset = None, 0  # sharding.ShardSpec / first pid of this rank's block
        self._encoder = None
        if self.enabled:
            self._load_meta_and_collection()
            self._init_searcher()

    @classmethod
    def from_config(cls, cfg) -> "ColBERTRetriever":
        key = _identity(cfg.retrieval) + (str(getattr(cfg.retrieval, "shard", None) or "none"),)
        with cls._registry_lock:
            if key not in cls._instances_by_key:
                cls._instances_by_key[key] = cls(cfg)
            return cls._instances_by_key[key]

    # pid -> chunk map, refreshed when the meta file changes on disk
    def _load_meta_and_collection(self) -> None:
        if not self.meta_file.exists():
            raise RuntimeError(f"ColBERT meta file not found: {self.meta_file}. Run build_colbert_index() first.")
        stamp = self.meta_file.stat().st_mtime
        if stamp == self._meta_mtime:
            return
        by_pid = artifacts.read_colbert_meta(self.meta_file)
        if not by_pid:
            raise RuntimeError(f"ColBERT meta file is empty: {self.meta_file}")
        texts = [""] * (max(by_pid) + 1)
        for pid, chunk in by_pid.items():
            texts[pid] = (chunk.text or "").strip()
        self._pid2chunk, self._collection, self._meta_mtime = by_pid, texts, stamp

    # token store -> HBM (one MaxSim index per index identity and process)
    def _init_searcher(self) -> None:
        rcfg = self.cfg.retrieval
        self._encoder = get_token_encoder(self.model_name, str(getattr(rcfg, "encoder_backend", "auto")),
                                          int(getattr(rcfg, "colbert_doc_maxlen", 220)),
                                          device=f"cuda:{self.device_index}")
        from . import sharding
        self.shard = sharding.active_shard(rcfg)
        key = (str(self.index_path), self.index_name, str(self.model_name), self.experiment, self.nranks,
               self.shard.key if self.shard else None)
        with type(self)._registry_lock:
            if key not in type(self)._searcher_cache:
                tokens, doc_ptr = artifacts.read_token_store(
                    artifacts.colbert_index_dir(str(self.index_path), self.experiment, self.index_name))
                lo = 0
                if self.shard is not None:
                    # row-sharded deployment: the token vectors of this rank's documents only (pid = lo + local id)
                    lo, hi = self.shard.bounds(len(doc_ptr) - 1)
                    tokens = np.ascontiguousarray(tokens[int(doc_ptr[lo]):int(doc_ptr[hi])])
                    doc_ptr = np.ascontiguousarray(doc_ptr[lo:hi + 1] - doc_ptr[lo])
                type(self)._searcher_cache[key] = (_native.MaxSimIndex(tokens, doc_ptr, device=self.device_index), lo)
            self._searcher, self.row_offset = type(self)._searcher_cache[key]

    def search(self, query: str, top_k: int = 5) -> List[Tuple[LawChunk, float]]:
        if not self.enabled:
            return []
        if self._searcher is None:
            raise RuntimeError("ColBERT Searcher is not initialized.")
        self._load_meta_and_collection()
        question = (query or "").strip()
        if not question:
            return []
        depth = max(1, int(top_k))
        q_tokens = np.asarray(self._encoder.encode_query(question), dtype=np.float32)[None]
        try:
            if self.shard is not None:
                if depth > _native.MAX_K:
                    raise ValueError(f"sharded ColBERT search: depth {depth} exceeds the kernels' limit of {_native.MAX_K}")
                from . import sharding
                ls, lp = self._searcher.search(q_tokens, depth)
                (scores, pids), = sharding.exchange_topk_numpy([(ls, lp)], self.row_offset, self.device_index,
                                                               group=self.shard.group)
            elif depth <= _native.MAX_K:
                scores, pids = self._searcher.search(q_tokens, depth)
            else:  # beyond the kernels' depth: every document's MaxSim score, stable sort on the host
                full = self._searcher.scores(q_tokens)[0]
                order = np.argsort(-full, kind="stable")[:depth]
                scores, pids = full[order][None], order.astype(np.int64)[None]
        except _native.NativeError as exc:
            if "out of memory" in str(exc).lower():
                return []  # the reference answers GPU OOM with an empty channel (:153-172)
            raise
        found = ((self._pid2chunk.get(int(p)), float(s)) for p, s in zip(pids[0], scores[0]) if p >= 0)
        return [(chunk, s) for chunk, s in found if chunk is not None]
