"""ColBERT channel (legalrag/retrieval/colbert_retriever.py:30-183).

Same contract: `ColBERTRetriever(cfg)` / `.from_config(cfg)` singletons,
`.enabled`, `search(query, top_k=5) -> [(LawChunk, score)]`, RuntimeError when
the meta file is missing, `[]` for an empty query, when disabled, or on GPU
out-of-memory.  The reference delegates to colbert-ai's PLAID Searcher; here the
query token matrix [32,128] is scored EXHAUSTIVELY against every document's
token embeddings with the MaxSim HIP kernel (the corpus shard fits HBM many
times over), so there is no centroid pruning and no residual quantisation.
pid == row of colbert_meta.jsonl, exactly as the reference builder writes it.
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


def get_token_encoder(model_name: Optional[str], backend: str, doc_maxlen: int):
    key = (str(model_name), backend, int(doc_maxlen))
    if key in _TOKEN_ENCODERS:
        return _TOKEN_ENCODERS[key]
    if backend == "hashing":
        enc = encoders.HashingTokenEmbedder(doc_maxlen=doc_maxlen)
    elif model_name and os.path.isdir(str(model_name)):
        enc = encoders.TransformersColBERT(str(model_name), doc_maxlen=doc_maxlen)
    else:
        enc = None
    if enc is not None:
        _TOKEN_ENCODERS[key] = enc
        return enc
    raise RuntimeError(
        f"ColBERT checkpoint '{model_name}' cannot be loaded offline (jina-colbert-v2 needs remote code and a "
        f"download); set cfg.retrieval.encoder_backend='hashing' for the deterministic stand-in token encoder.")


class ColBERTRetriever:
    _instances_by_key: ClassVar[Dict[Tuple[str, str, str, str, str, int], "ColBERTRetriever"]] = {}
    _searcher_cache: ClassVar[Dict[Tuple[str, str, str, str, int], object]] = {}
    _lock: ClassVar[threading.Lock] = threading.Lock()

    def __init__(self, cfg):
        self.cfg = cfg
        rcfg = cfg.retrieval
        self.enabled: bool = bool(getattr(rcfg, "enable_colbert", False))
        self.index_path: Path = Path(str(getattr(rcfg, "colbert_index_path")))
        self.index_name: str = str(getattr(rcfg, "colbert_index_name"))
        self.model_name: Optional[str] = getattr(rcfg, "colbert_model_name", "colbert-ir/colbertv2.0")
        self.meta_file: Path = Path(str(getattr(rcfg, "colbert_meta_file")))
        self.experiment: str = str(getattr(rcfg, "colbert_experiment"))
        self.nranks: int = int(getattr(rcfg, "colbert_nranks", 1))
        self.device_index = int(getattr(rcfg, "device", 0))
        self._pid2chunk: Dict[int, LawChunk] = {}
        self._collection: List[str] = []
        self._searcher: Optional[_native.MaxSimIndex] = None
        self._encoder = None
        self._meta_mtime: float | None = None
        if not self.enabled:
            return
        self._load_meta_and_collection()
        self._init_searcher()

    @classmethod
    def from_config(cls, cfg) -> "ColBERTRetriever":
        rcfg = cfg.retrieval
        key = (str(getattr(rcfg, "colbert_index_path")), str(getattr(rcfg, "colbert_index_name")),
               str(getattr(rcfg, "colbert_model_name", "colbert-ir/colbertv2.0")),
               str(getattr(rcfg, "colbert_meta_file")), str(getattr(rcfg, "colbert_experiment")),
               int(getattr(rcfg, "colbert_nranks", 1)))
        with cls._lock:
            inst = cls._instances_by_key.get(key)
            if inst is None:
                inst = cls(cfg)
                cls._instances_by_key[key] = inst
            return inst

    def _load_meta_and_collection(self) -> None:
        if not self.meta_file.exists():
            raise RuntimeError(f"ColBERT meta file not found: {self.meta_file}. Run build_colbert_index() first.")
        meta_mtime = self.meta_file.stat().st_mtime
        if self._meta_mtime is not None and self._meta_mtime == meta_mtime:
            return
        pid2chunk = artifacts.read_colbert_meta(self.meta_file)
        if not pid2chunk:
            raise RuntimeError(f"ColBERT meta file is empty: {self.meta_file}")
        collection: List[str] = [""] * (max(pid2chunk) + 1)
        for pid, chunk in pid2chunk.items():
            collection[pid] = (getattr(chunk, "text", "") or "").strip()
        self._pid2chunk = pid2chunk
        self._collection = collection
        self._meta_mtime = meta_mtime

    def _init_searcher(self) -> None:
        key = (str(self.index_path), str(self.index_name), str(self.model_name), str(self.experiment), int(self.nranks))
        rcfg = self.cfg.retrieval
        self._encoder = get_token_encoder(self.model_name, str(getattr(rcfg, "encoder_backend", "auto")),
                                          int(getattr(rcfg, "colbert_doc_maxlen", 220)))
        cached = self.__class__._searcher_cache.get(key)
        if cached is not None:
            self._searcher = cached
            return
        D, doc_ptr = artifacts.read_token_store(
            artifacts.colbert_index_dir(str(self.index_path), self.experiment, self.index_name))
        self._searcher = _native.MaxSimIndex(D, doc_ptr, device=self.device_index)
        self.__class__._searcher_cache[key] = self._searcher

    def search(self, query: str, top_k: int = 5) -> List[Tuple[LawChunk, float]]:
        if not self.enabled:
            return []
        if not self._searcher:
            raise RuntimeError("ColBERT Searcher is not initialized.")
        self._load_meta_and_collection()
        query = (query or "").strip()
        if not query:
            return []
        k = max(1, min(int(top_k), _native.MAX_K))
        try:
            Q = np.asarray(self._encoder.encode_query(query), dtype=np.float32)[None]
            scores, pids = self._searcher.search(Q, k)
        except _native.NativeError as exc:
            if "out of memory" in str(exc).lower():
                return []
            raise
        out: List[Tuple[LawChunk, float]] = []
        for pid, score in zip(pids[0].tolist(), scores[0].tolist()):
            chunk = self._pid2chunk.get(int(pid))
            if chunk is not None:
                out.append((chunk, float(score)))
        return out
