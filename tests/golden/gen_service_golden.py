#!/usr/bin/env python3
"""Golden vectors for the `/retrieve` service contract, produced by running the REFERENCE'S OWN
route function (legalrag/services/retrieval_api.py:51-77) with a fake retriever and a fake router
in place of its module globals.  Runs only in the build container (reference mounted read-only);
absent third-party wheels are empty placeholder modules, exactly as in gen_fusion_golden.py — the
route touches none of them.  Only inputs + outputs (JSON) are committed.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_service_golden.py
"""
from __future__ import annotations

import json
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")


class _Placeholder(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (Exception,), {})


for name in ("jieba", "rank_bm25", "faiss", "FlagEmbedding", "colbert", "colbert.infra", "openai"):
    sys.modules.setdefault(name, _Placeholder(name))

from fastapi import HTTPException  # noqa: E402
from legalrag.config import AppConfig  # noqa: E402
from legalrag.schemas import IssueType, LawChunk, RetrievalHit, RoutingDecision, RoutingMode, TaskType  # noqa: E402
from legalrag.services import retrieval_api as api  # noqa: E402

OUT = Path(__file__).resolve().parent


def mk_hit(i: int, score: float, rank: int) -> RetrievalHit:
    c = LawChunk(id=f"src.txt::{i}", law_name="Synthetic Code", article_no=f"§ {i}", article_id=str(i),
                 text=f"text of provision {i}", lang="en", source="src.txt")
    return RetrievalHit(chunk=c, score=score, rank=rank, source="retriever",
                        score_breakdown={"channel": ["dense"], "dense_raw": score})


class FakeRetriever:
    def __init__(self):
        self.calls = []

    def search(self, question, top_k=10, decision=None):
        self.calls.append({"question": question, "top_k": top_k, "mode": decision.mode.value})
        return [mk_hit(i, 1.0 - 0.01 * i, i + 1) for i in range(min(top_k, 4))]


class FakeRouter:
    def __init__(self, factor, mode):
        self.factor, self.mode = factor, mode

    def route(self, question):
        return RoutingDecision(task_type=list(TaskType)[0], issue_type=list(IssueType)[0], mode=self.mode,
                               top_k_factor=self.factor)


def main():
    cfg = AppConfig()
    cases = []
    bodies = [
        ({"question": "  what is a merchant?  "}, 1.0, RoutingMode.RAG),
        ({"question": "q", "top_k": 7}, 1.0, RoutingMode.RAG),
        ({"question": "q", "top_k": "12"}, 1.5, RoutingMode.GRAPH_AUGMENTED),
        ({"question": "q", "top_k": "not a number"}, 1.0, RoutingMode.RAG),
        ({"question": "q", "top_k": 1}, 1.0, RoutingMode.RAG),          # clamped up to 3
        ({"question": "q", "top_k": 25}, 2.0, RoutingMode.RAG),         # clamped down to 30
        ({"question": "q", "top_k": 5}, 0.5, RoutingMode.RAG),          # int(2.5) = 2 -> 3
        ({"question": "q", "top_k": 9}, 1.3, RoutingMode.RAG),          # int(11.7) = 11
        ({"question": "q", "top_k": None}, 1.2, RoutingMode.RAG),
        ({"question": "   "}, 1.0, RoutingMode.RAG),                    # 400
        ({}, 1.0, RoutingMode.RAG),                                      # 400
    ]
    for body, factor, mode in bodies:
        api.CFG, api.RETRIEVER, api.ROUTER = cfg, FakeRetriever(), FakeRouter(factor, mode)
        try:
            resp = api.retrieve(dict(body))
            cases.append({"body": body, "top_k_factor": factor, "mode": mode.value, "cfg_top_k": cfg.retrieval.top_k,
                          "response": resp, "search_calls": api.RETRIEVER.calls})
        except HTTPException as e:
            cases.append({"body": body, "top_k_factor": factor, "mode": mode.value, "cfg_top_k": cfg.retrieval.top_k,
                          "http_error": {"status_code": e.status_code, "detail": e.detail}})
    api.CFG = api.RETRIEVER = api.ROUTER = None
    try:
        api.retrieve({"question": "q"})
    except HTTPException as e:
        cases.append({"body": {"question": "q"}, "not_ready": True,
                      "http_error": {"status_code": e.status_code, "detail": e.detail}})
    (OUT / "service_golden.json").write_text(json.dumps({"cases": cases}, ensure_ascii=False, indent=1, default=str))
    print(f"wrote {len(cases)} cases")


if __name__ == "__main__":
    main()
