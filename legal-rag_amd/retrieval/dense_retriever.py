"""Dense channel (counterpart of legalrag/retrieval/dense_retriever.py:14-60).

`search(query, top_k)`: reload the store if its files changed, embed the query
(with the retrieval instruction), run the exact inner-product top-k on the GPU
and wrap the rows as RetrievalHits.  As in the reference, `rank` is the 1-based
position in the index result — positions whose id is out of range (the -1
padding when k > ntotal) are skipped and leave a gap — and `semantic_score`
repeats the score."""
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
        self.store = self.store or VectorStore.from_config(self.cfg)

    def search(self, query: str, top_k: int) -> List[RetrievalHit]:
        store = self.store
        store.load()
        depth = max(1, int(top_k))
        scores, rows = store.index.search(store._embed([query], is_query=True), depth)
        n_chunks = len(store.chunks)
        out: List[RetrievalHit] = []
        for position, (row, score) in enumerate(zip(rows[0].tolist(), scores[0].tolist()), start=1):
            if 0 <= row < n_chunks:
                s = float(score)
                out.append(RetrievalHit(chunk=store.chunks[row], score=s, rank=position, source="retriever",
                                        semantic_score=s))
        return out
