"""Language routing (counterpart of legalrag/retrieval/by_lang_retriever.py:11-29).

One HybridRetriever per query language: the zh retriever is built with the
config handed in (eagerly, like the reference — server start-up warms it), any
other language on its first query from `cfg.with_lang(lang)`.  Creation is
guarded by a lock because the /retrieve service calls `search` from a thread
pool."""
from __future__ import annotations

import threading
from typing import Any, Dict

from ..text import detect_lang
from .hybrid_retriever import HybridRetriever


class ByLangRetriever:
    def __init__(self, cfg):
        self._base_cfg = cfg
        self._guard = threading.Lock()
        self._retriever_cfgs: Dict[str, Any] = {"zh": cfg}
        self._retrievers: Dict[str, HybridRetriever] = {"zh": HybridRetriever(cfg)}

    def _for_lang(self, lang: str) -> HybridRetriever:
        found = self._retrievers.get(lang)
        if found is not None:
            return found
        with self._guard:
            if lang not in self._retrievers:
                self._retriever_cfgs[lang] = self._base_cfg.with_lang(lang)
                self._retrievers[lang] = HybridRetriever(self._retriever_cfgs[lang])
            return self._retrievers[lang]

    def search(self, question: str, llm=None, top_k: int = 10, decision=None):
        return self._for_lang(detect_lang(question)).search(question, llm=llm, top_k=top_k, decision=decision)
