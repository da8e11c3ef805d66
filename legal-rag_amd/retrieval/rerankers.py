"""Rerankers (legalrag/retrieval/rerankers.py): cross-encoder / LLM scoring of
(query, candidate) pairs and score normalisation.

Same public surface: RerankResult, minmax_normalize, sigmoid, sigmoid_calibrate,
CrossEncoderReranker.score/score_batch, LLMReranker (+ cached / async forms),
RerankerFactory(llm, cross_model, llm_threshold, use_cache).create(top_k),
rerank_candidates(query, candidates, reranker, *, top_n, content_key,
normalize, sigmoid_temperature, include_debug).  The model forward stays in
PyTorch-ROCm (plain transformers; sentence-transformers is absent here); the
blend of the normalised scores into the fused ranking runs on the GPU
(amdr_rerank_blend, see hybrid_retriever.py).
"""
from __future__ import annotations

import asyncio
import json
import math
import os
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Protocol, Sequence, Tuple, Union

TextLike = Union[str, Dict[str, Any]]


class BaseReranker(Protocol):
    def score(self, query: str, doc: str) -> float: ...

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        return [self.score(query, d) for d in docs]


@dataclass(frozen=True)
class RerankResult:
    raw_score: float
    norm_score: float
    meta: Dict[str, Any]


def minmax_normalize(scores: Sequence[float]) -> List[float]:
    if not scores:
        return []
    lo, hi = min(scores), max(scores)
    if hi - lo < 1e-12:
        return [0.0] * len(scores)
    return [(s - lo) / (hi - lo) for s in scores]


def sigmoid(x: float) -> float:
    if x >= 0:
        z = math.exp(-x)
        return 1.0 / (1.0 + z)
    z = math.exp(x)
    return z / (1.0 + z)


def sigmoid_calibrate(scores: Sequence[float], temperature: float = 1.0) -> List[float]:
    t = max(1e-6, float(temperature))
    return [sigmoid(s / t) for s in scores]


def _safe_clip(x: float, lo: float, hi: float) -> float:
    return max(lo, min(hi, x))


def _to_doc_text(doc: TextLike, content_key: str = "text") -> str:
    """rerankers.py:78-86 — NOTE: anything that is neither str nor dict (e.g. the
    RetrievalHit objects HybridRetriever.search passes) becomes str(doc)."""
    if isinstance(doc, str):
        return doc
    if isinstance(doc, dict):
        for k in (content_key, "text", "content", "provision", "chunk", "body"):
            if k in doc and isinstance(doc[k], str):
                return doc[k]
        return str(doc)
    return str(doc)


@dataclass
class CrossEncoderReranker:
    """Sequence-classification cross-encoder on PyTorch-ROCm.  Mirrors
    sentence_transformers.CrossEncoder.predict for 1-label models: sigmoid of the
    logit, batches of `batch_size`, pairs truncated to `max_length`."""
    model_name: str = "BAAI/bge-reranker-base"
    device: Optional[str] = None
    max_length: int = 512
    batch_size: int = 32

    def __post_init__(self):
        if not os.path.isdir(self.model_name):
            raise RuntimeError(
                f"cross-encoder '{self.model_name}' is not a local checkpoint directory and this build has no "
                f"network; point cfg.retrieval.rerank_ce_model at a downloaded checkpoint.")
        import torch
        from transformers import AutoModelForSequenceClassification, AutoTokenizer
        self._torch = torch
        self._device = torch.device(self.device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self._tok = AutoTokenizer.from_pretrained(self.model_name, local_files_only=True)
        self._model = AutoModelForSequenceClassification.from_pretrained(
            self.model_name, local_files_only=True).to(self._device).eval()
        if self._device.type == "cuda":
            self._model = self._model.half()

    def score(self, query: str, doc: str) -> float:
        return self.score_batch(query, [doc])[0]

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        torch = self._torch
        out: List[float] = []
        with torch.inference_mode():
            for s in range(0, len(docs), self.batch_size):
                batch = docs[s:s + self.batch_size]
                enc = self._tok([query] * len(batch), batch, padding=True, truncation=True,
                                max_length=self.max_length, return_tensors="pt").to(self._device)
                logits = self._model(**enc).logits.float()
                if logits.shape[-1] == 1:
                    logits = torch.sigmoid(logits[:, 0])
                else:
                    logits = logits[:, -1]
                out.extend(float(x) for x in logits.cpu())
        return out


LLM_RERANK_SYSTEM_PROMPT = (
    "You are a precise ranking model. Evaluate how well a candidate legal provision answers a user query. "
    'Output ONLY a JSON object {"score": float, "reason": "string"} with score between 0 and 1 '
    "(1.0 = directly answers the query, 0.0 = irrelevant); keep the reason short."
)


def build_llm_rerank_prompt(query: str, provision: str) -> str:
    return f"\nQuery:\n{query}\n\nCandidate provision:\n{provision}\n\nEvaluate relevance and return JSON only.\n"


@dataclass
class LLMReranker:
    llm: Any
    temperature: float = 0.0
    max_query_chars: int = 800
    max_doc_chars: int = 2000

    def _truncate(self, s: str, n: int) -> str:
        return s if len(s) <= n else s[:n] + "…"

    def _call_llm(self, query: str, doc: str) -> str:
        messages = [{"role": "system", "content": LLM_RERANK_SYSTEM_PROMPT},
                    {"role": "user", "content": build_llm_rerank_prompt(self._truncate(query, self.max_query_chars),
                                                                         self._truncate(doc, self.max_doc_chars))}]
        return str(self.llm.chat(messages=messages, tag="rerank_llm"))

    def score(self, query: str, doc: str) -> float:
        return _safe_clip(self._extract_score(self._call_llm(query, doc)), 0.0, 1.0)

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        return [self.score(query, d) for d in docs]

    @staticmethod
    def _extract_score(text: str) -> float:
        t = (text or "").strip()
        try:
            obj = json.loads(t)
            if isinstance(obj, dict) and "score" in obj:
                return float(obj["score"])
        except Exception:  # noqa: BLE001
            pass
        m = re.search(r"([0-1](?:\.\d+)?)", t)
        return float(m.group(1)) if m else 0.0


@dataclass
class AsyncLLMReranker:
    llm: Any
    max_concurrency: int = 8
    base: Optional[LLMReranker] = None

    def __post_init__(self):
        if self.base is None:
            self.base = LLMReranker(llm=self.llm)

    async def _call_llm_async(self, query: str, doc: str) -> str:
        if hasattr(self.llm, "achat"):
            messages = [{"role": "system", "content": LLM_RERANK_SYSTEM_PROMPT},
                        {"role": "user", "content": build_llm_rerank_prompt(query, doc)}]
            return str(await self.llm.achat(messages=messages, tag="rerank_llm"))
        loop = asyncio.get_running_loop()
        return await loop.run_in_executor(None, lambda: self.base.score(query, doc))

    async def score(self, query: str, doc: str) -> float:
        text = await self._call_llm_async(query, doc)
        return _safe_clip(self.base._extract_score(str(text)), 0.0, 1.0)

    async def score_batch(self, query: str, docs: List[str]) -> List[float]:
        sem = asyncio.Semaphore(self.max_concurrency)

        async def _one(d):
            async with sem:
                return await self.score(query, d)
        return list(await asyncio.gather(*[_one(d) for d in docs]))


@dataclass
class CachedLLMReranker(LLMReranker):
    cache: Dict[Tuple[int, int], float] = None

    def __post_init__(self):
        if self.cache is None:
            self.cache = {}

    def score(self, query: str, doc: str) -> float:
        key = (hash(query), hash(doc))
        if key in self.cache:
            return self.cache[key]
        s = super().score(query, doc)
        self.cache[key] = s
        return s


@dataclass
class AsyncCachedLLMReranker(AsyncLLMReranker):
    cache: Dict[Tuple[int, int], float] = None

    def __post_init__(self):
        super().__post_init__()
        if self.cache is None:
            self.cache = {}

    async def score(self, query: str, doc: str) -> float:
        key = (hash(query), hash(doc))
        if key in self.cache:
            return self.cache[key]
        s = await super().score(query, doc)
        self.cache[key] = s
        return s


class RerankerFactory:
    """LLM reranker when an llm is given and top_k <= llm_threshold, else the
    (class-level cached) cross-encoder (rerankers.py:281-312)."""
    _cross_cache: Dict[str, Any] = {}

    def __init__(self, llm: Any = None, cross_model: str = "BAAI/bge-reranker-base", llm_threshold: int = 30,
                 use_cache: bool = True):
        self.llm = llm
        self.cross_model = cross_model
        self.llm_threshold = llm_threshold
        self.use_cache = use_cache
        self._cache: Dict[Tuple[int, int], float] = {}

    def create(self, top_k: int):
        if self.llm is not None and top_k <= self.llm_threshold:
            if self.use_cache:
                return CachedLLMReranker(llm=self.llm, cache=self._cache)
            return LLMReranker(llm=self.llm)
        cache = self.__class__._cross_cache
        if self.cross_model in cache:
            return cache[self.cross_model]
        reranker = CrossEncoderReranker(model_name=self.cross_model)
        cache[self.cross_model] = reranker
        return reranker


def rerank_candidates(query: str, candidates: Sequence[TextLike], reranker: BaseReranker, *, top_n: int,
                      content_key: str = "text", normalize: str = "minmax", sigmoid_temperature: float = 1.0,
                      include_debug: bool = False) -> List[Tuple[TextLike, RerankResult]]:
    if top_n <= 0:
        return []
    docs = [_to_doc_text(c, content_key=content_key) for c in candidates]
    raw = reranker.score_batch(query, docs)
    if normalize == "none":
        norm = list(raw)
    elif normalize == "sigmoid":
        norm = sigmoid_calibrate(raw, temperature=sigmoid_temperature)
    else:
        norm = minmax_normalize(raw)
    results = []
    for c, rs, ns in zip(candidates, raw, norm):
        meta = {"raw": rs, "norm": ns} if include_debug else {}
        results.append((c, RerankResult(raw_score=rs, norm_score=ns, meta=meta)))
    results.sort(key=lambda x: x[1].norm_score, reverse=True)
    return results[:top_n]
