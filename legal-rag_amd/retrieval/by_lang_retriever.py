"""Language routing (legalrag/retrieval/by_lang_retriever.py:11-29): one
HybridRetriever per detected query language, the zh one built eagerly and any
other on first use via `cfg.with_lang(lang)`."""
from __future__ import annotations

from typing import Any, Dict

from ..text import detect_lang
from .hybrid_retriever import HybridRetriever


class ByLangRetriever:
    def __init__(self, cfg):
        self._base_cfg = cfg
        self._retrievers: Dict[str, HybridRetriever] = {"zh": HybridRetriever(cfg)}
        self._retriever_cfgs: Dict[str, Any] = {"zh": cfg}

    def search(self, question: str, llm=None, top_k: int = 10, decision=None):
        lang = detect_lang(question)
        retriever = self._retrievers.get(lang)
        if retriever is None:
            lang_cfg = self._base_cfg.with_lang(lang)
            retriever = HybridRetriever(lang_cfg)
            self._retrievers[lang] = retriever
            self._retriever_cfgs[lang] = lang_cfg
        return retriever.search(question, llm=llm, top_k=top_k, decision=decision)
