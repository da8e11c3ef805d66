"""Dense channel (legalrag/retrieval/dense_retriever.py:14-60): embed the query,
exact inner-product top-k on the GPU, wrap RetrievalHits.  Ranks are the 1-based
positions in the index result with gaps kept when an id is dropped (:46-48)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

from ..schemas import RetrievalHit
from .vector_store import VectorStore


@dataclass
class DenseRetriever:
    cfg: Any
    store: Optional[VectorStore] = None

    def __post_init__(self) -> None:
        if self.store is None:
            self.store = VectorStore.from_config(self.cfg)

    def search(self, query: str, top_k: int) -> List[RetrievalHit]:
        assert self.store is not None
        self.store.load()
        k = max(1, int(top_k))
        q_vec = self.store._embed([query], is_query=True)
        scores, idxs = self.store.index.search(q_vec, k)
        scores = scores[0].tolist()
        idxs = idxs[0].tolist()
        hits: List[RetrievalHit] = []
        for rank, (i, s) in enumerate(zip(idxs, scores), start=1):
            if i < 0 or i >= len(self.store.chunks):
                continue
            hits.append(RetrievalHit(chunk=self.store.chunks[i], score=float(s), rank=rank, source="retriever",
                                     semantic_score=float(s)))
        return hits
