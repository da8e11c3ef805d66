"""Data objects returned to callers of the retrieval path.

Field-for-field mirror of legalrag/schemas.py:9-32 (LawChunk, RetrievalHit):
RagPipeline and the HTTP services serialise these, and the reference's rerank
stage feeds `str(hit)` — the pydantic repr — to the cross-encoder
(hybrid_retriever.py:343 + rerankers.py:78-86), so class names, field names,
field order and defaults must all match for results to be identical.
"""
from __future__ import annotations

from typing import Any, Dict, List, Literal, Optional

from pydantic import BaseModel, ConfigDict


class LawChunk(BaseModel):
    id: str
    law_name: str
    chapter: Optional[str] = None
    section: Optional[str] = None
    article_no: str
    article_id: str
    text: str
    lang: Optional[str] = "zh"
    source: Optional[str] = None
    start_char: Optional[int] = None
    end_char: Optional[int] = None


class RetrievalHit(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    chunk: LawChunk
    score: float
    rank: Optional[int] = None
    source: Literal["retriever", "graph", "rerank"] = "retriever"
    semantic_score: Optional[float] = None
    graph_depth: Optional[int] = None
    relations: Optional[List[str]] = None
    seed_article_id: Optional[str] = None
    score_breakdown: Optional[Dict[str, Any]] = None
