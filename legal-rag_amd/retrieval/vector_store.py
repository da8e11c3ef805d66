"""Dense vector backend: encoder host + exact inner-product index in HBM.

Mirror of legalrag/retrieval/vector_store.py (same attributes `.index .chunks
.index_path .meta_path .model`, same method contracts, same error types):
  * `VectorStore(cfg)` / `from_config(cfg)`  — process-wide singleton per
    (model, index file, meta file, device)                 (:79-93)
  * `load()`     — mtime-guarded reload, FileNotFoundError if either file is
    missing                                                 (:95-128)
  * `_embed(texts, is_query)` -> float32 [n, d]; empty list -> zeros((0, d)) (:131-155)
  * `search(query, top_k)` -> [(LawChunk, float)]           (:157-180)
`self.index` is a faiss-shaped object (`search(q, k)`, `add(x)`, `ntotal`, `d`)
backed by the HIP scan kernel instead of IndexHNSWFlat; HNSW knobs are accepted
and ignored because the scan is exact.
"""
from __future__ import annotations

import threading
from pathlib import Path
from typing import ClassVar, Dict, List, Tuple

import numpy as np

from .. import _native, artifacts, encoders
from ..schemas import LawChunk


class FlatIPIndex:
    """faiss.Index-shaped facade over _native.DenseIndex."""

    def __init__(self, X: np.ndarray, device: int = 0):
        self._idx = _native.DenseIndex(X, device=device)
        self.d = int(X.shape[1])
        self.metric_type = artifacts.METRIC_INNER_PRODUCT

    @property
    def ntotal(self) -> int:
        return self._idx.ntotal

    FULL_SCORES_MAX_ROWS = 1 << 20  # k beyond the kernels' depth limit: score every row, sort on the host

    def search(self, q: np.ndarray, k: int):
        """faiss `index.search` contract for any k (the reference accepts any cfg.retrieval.top_k):
        up to AMDR_MAX_K the fused top-k kernels; deeper requests score every row on the device
        (amdr_dense_score_rows) and take a stable descending sort on the host, like the BM25 channel."""
        k = int(k)
        if k <= _native.MAX_K:
            return self._idx.search(q, k)
        n = self.ntotal
        if n > self.FULL_SCORES_MAX_ROWS:
            raise ValueError(f"dense search depth {k} exceeds the kernels' limit of {_native.MAX_K} and the index has "
                             f"{n} rows (full-score path is limited to {self.FULL_SCORES_MAX_ROWS})")
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.d)
        rows = np.tile(np.arange(n, dtype=np.int64), (q.shape[0], 1))
        full = self._idx.score_rows(q, rows) if n else np.zeros((q.shape[0], 0), dtype=np.float32)
        scores = np.full((q.shape[0], k), -np.finfo(np.float32).max, dtype=np.float32)
        ids = np.full((q.shape[0], k), -1, dtype=np.int64)
        for b in range(q.shape[0]):
            order = np.argsort(-full[b], kind="stable")[:k]
            scores[b, :len(order)] = full[b, order]
            ids[b, :len(order)] = order
        return scores, ids

    def add(self, x: np.ndarray) -> None:
        self._idx.add(x)

    def reconstruct_n(self, i0: int, n: int) -> np.ndarray:
        return self._idx.read_rows(int(i0), int(n))

    @property
    def native(self) -> "_native.DenseIndex":
        return self._idx


class ShardedFlatIPIndex(FlatIPIndex):
    """This rank's row block [lo, hi) of the chunk matrix in HBM behind the same faiss-shaped facade: `search`
    returns GLOBAL row ids of the global top-k — local scan, one all-gather of the per-shard lists, merge
    (retrieval/sharding.py) — identical on every rank.  `ntotal` is the global row count; `native` is the local
    index (rows lo .. hi-1 at local ids 0 .. hi-lo-1, `row_offset` = lo)."""

    def __init__(self, X: np.ndarray, spec, device: int = 0):
        self.spec = spec
        self.n_global = int(X.shape[0])
        self.row_offset, self.row_end = spec.bounds(self.n_global)
        super().__init__(np.ascontiguousarray(X[self.row_offset:self.row_end]), device=device)
        self._device = int(device)

    @property
    def ntotal(self) -> int:
        return self.n_global

    def search(self, q: np.ndarray, k: int):
        from . import sharding
        k = int(k)
        if k > _native.MAX_K:
            raise ValueError(f"sharded dense search: depth {k} exceeds the kernels' limit of {_native.MAX_K}")
        s, i = self._idx.search(q, k)
        (gs, gi), = sharding.exchange_topk_numpy([(s, i)], self.row_offset, self._device, group=self.spec.group)
        return gs, gi

    def add(self, x: np.ndarray) -> None:
        raise RuntimeError("incremental add on a row-sharded index is not supported: rebuild and reload")

    def reconstruct_n(self, i0: int, n: int) -> np.ndarray:
        """rows by GLOBAL id; only this rank's block is resident"""
        i0, n = int(i0), int(n)
        if i0 < self.row_offset or i0 + n > self.row_end:
            raise IndexError(f"rows [{i0}, {i0 + n}) are not in this rank's shard [{self.row_offset}, {self.row_end})")
        return self._idx.read_rows(i0 - self.row_offset, n)


class VectorStore:
    _instances_by_key: ClassVar[Dict[Tuple[str, ...], "VectorStore"]] = {}
    _lock: ClassVar[threading.Lock] = threading.Lock()

    def __init__(self, cfg):
        self.cfg = cfg
        rcfg = cfg.retrieval
        self.index_path = Path(rcfg.faiss_index_file)
        self.meta_path = Path(rcfg.faiss_meta_file)
        self.device_index = int(getattr(rcfg, "device", 0))
        self.device = f"cuda:{self.device_index}"
        self.model = encoders.get_embedder(str(rcfg.embedding_model),
                                           backend=str(getattr(rcfg, "encoder_backend", "auto")),
                                           dim=int(getattr(rcfg, "embedding_dim", 768)), device=self.device)
        self.index: FlatIPIndex | None = None
        self.chunks: List[LawChunk] = []
        self._index_mtime: float | None = None
        self._meta_mtime: float | None = None
        self._load_lock = threading.Lock()

    @classmethod
    def from_config(cls, cfg) -> "VectorStore":
        rcfg = cfg.retrieval
        key = (str(rcfg.embedding_model), str(rcfg.faiss_index_file), str(rcfg.faiss_meta_file),
               f"cuda:{int(getattr(rcfg, 'device', 0))}", str(getattr(rcfg, "shard", None) or "none"))
        with cls._lock:
            inst = cls._instances_by_key.get(key)
            if inst is None:
                inst = cls(cfg)
                cls._instances_by_key[key] = inst
            return inst

    def load(self) -> None:
        if not self.index_path.exists() or not self.meta_path.exists():
            raise FileNotFoundError("FAISS 索引或元数据不存在，请先运行 scripts.build_index.")
        index_mtime = self.index_path.stat().st_mtime
        meta_mtime = self.meta_path.stat().st_mtime
        if (self.index is not None and self.chunks and self._index_mtime == index_mtime
                and self._meta_mtime == meta_mtime):
            return
        with self._load_lock:  # concurrent searches from the service thread pool
            if (self.index is not None and self.chunks and self._index_mtime == index_mtime
                    and self._meta_mtime == meta_mtime):
                return
            X, _metric = artifacts.read_faiss_index(self.index_path)
            chunks = artifacts.read_faiss_meta(self.meta_path)
            from . import sharding
            spec = sharding.active_shard(self.cfg.retrieval)
            # row-sharded deployment: only this rank's row block goes to HBM; the chunk list (host) stays whole,
            # global row id == position in it
            self.index = (ShardedFlatIPIndex(X, spec, device=self.device_index) if spec is not None
                          else FlatIPIndex(X, device=self.device_index))
            self.chunks = chunks
            self._index_mtime = index_mtime
            self._meta_mtime = meta_mtime

    def _embed(self, texts, is_query: bool = False) -> np.ndarray:
        if isinstance(texts, str):  # graph_retriever.py:178 calls with a bare str
            texts = [texts]
            single = True
        else:
            single = False
        if not texts:
            return np.zeros((0, int(self.model.hidden_size)), dtype="float32")
        if is_query:
            embs = self.model.encode_queries(list(texts), batch_size=64, max_length=512)
        else:
            embs = self.model.encode(list(texts), batch_size=64, max_length=512)
        embs = np.asarray(embs).astype("float32")
        return embs[0] if single else embs

    def embed_device(self, texts: List[str], is_query: bool = False):
        """Embeddings as a contiguous fp32 torch tensor on this store's device.  A PyTorch encoder
        (TransformersBGE) hands its output over without leaving HBM; the hashing stand-in is
        computed on the host and uploaded."""
        import torch
        tdev = torch.device("cuda", self.device_index)
        if hasattr(self.model, "encode_tensor"):
            t = self.model.encode_tensor(list(texts), batch_size=64, max_length=512, is_query=is_query)
            return t.to(tdev).contiguous()
        return torch.from_numpy(np.ascontiguousarray(self._embed(list(texts), is_query=is_query), dtype=np.float32)) \
            .to(tdev, non_blocking=True)

    def search(self, query: str, top_k: int) -> List[Tuple[LawChunk, float]]:
        self.load()
        q_vec = self._embed([query], is_query=True)
        scores, idxs = self.index.search(q_vec, top_k)
        hits: List[Tuple[LawChunk, float]] = []
        for score, idx in zip(scores[0], idxs[0]):
            if idx == -1:
                continue
            hits.append((self.chunks[idx], float(score)))
        return hits
