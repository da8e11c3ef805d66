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


class Neighbor(BaseModel):
    """Directed edge of the law graph (legalrag/schemas.py:129-134)."""
    article_id: str
    relation: str = "neighbor"
    conf: float = 1.0
    evidence: Optional[Dict[str, Any]] = None


class LawNode(BaseModel):
    """Node of the law graph (legalrag/schemas.py:136-150).  The last three fields are filled
    per query on a COPY of the stored node by LawGraphStore.walk; `relations` is declared a
    string upstream but walk stores a one-element list in it (pydantic does not validate
    assignments), so it is typed loosely here."""
    article_id: str
    article_no: str = ""
    law_name: Optional[str] = None
    title: Optional[str] = None
    chapter: Optional[str] = None
    section: Optional[str] = None
    neighbors: List[Neighbor] = []
    meta: Dict[str, Any] = {}
    graph_depth: Optional[int] = None
    graph_parent: Optional[str] = None
    relations: Optional[Any] = None
