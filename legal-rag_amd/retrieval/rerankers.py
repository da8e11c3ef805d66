"""Rerankers: scoring of (query, candidate) pairs and score normalisation.

Public surface of legalrag/retrieval/rerankers.py kept name for name —
RerankResult, minmax_normalize, sigmoid, sigmoid_calibrate,
CrossEncoderReranker(.score/.score_batch), LLMReranker, AsyncLLMReranker,
CachedLLMReranker, AsyncCachedLLMReranker, RerankerFactory(llm, cross_model,
llm_threshold, use_cache).create(top_k), rerank_candidates(query, candidates,
reranker, *, top_n, content_key, normalize, sigmoid_temperature, include_debug).

What runs where: the cross-encoder forward is PyTorch-ROCm (plain transformers;
sentence-transformers is absent here); LLM rerankers only format a prompt and
parse a number; blending the normalised scores into the fused ranking is the
HIP kernel amdr_rerank_blend (see hybrid_retriever.py).
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
_DOC_KEYS = ("text", "content", "provision", "chunk", "body")


# --------------------------------------------------------------------- results
@dataclass(frozen=True)
class RerankResult:
    raw_score: float
    norm_score: float
    meta: Dict[str, Any]


class BaseReranker(Protocol):
    def score(self, query: str, doc: str) -> float: ...

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        return [self.score(query, d) for d in docs]


# ---------------------------------------------------------------- normalisers
def minmax_normalize(scores: Sequence[float]) -> List[float]:
    """(s - min) / (max - min); a flat or empty list maps to zeros (rerankers.py:48-54)."""
    n = len(scores)
    if n == 0:
        return []
    lo, hi = min(scores), max(scores)
    span = hi - lo
    if span < 1e-12:
        return [0.0] * n
    return [(s - lo) / span for s in scores]


def sigmoid(x: float) -> float:
    """Overflow-safe logistic: the branch keeps exp()'s argument non-positive (rerankers.py:57-62)."""
    e = math.exp(-abs(x))
    return 1.0 / (1.0 + e) if x >= 0 else e / (1.0 + e)


def sigmoid_calibrate(scores: Sequence[float], temperature: float = 1.0) -> List[float]:
    t = max(1e-6, float(temperature))
    return [sigmoid(s / t) for s in scores]


def _clip01(x: float) -> float:
    return min(1.0, max(0.0, x))


def _to_doc_text(doc: Any, content_key: str = "text") -> str:
    """Text handed to the reranker (rerankers.py:78-86).  Strings pass through, dicts give
    their first string field among content_key/text/content/provision/chunk/body — and
    ANYTHING ELSE, including the RetrievalHit objects HybridRetriever.search passes,
    becomes str(doc): the cross-encoder really does read the pydantic repr of the hit."""
    if isinstance(doc, str):
        return doc
    if isinstance(doc, dict):
        for key in (content_key,) + _DOC_KEYS:
            val = doc.get(key)
            if isinstance(val, str):
                return val
    return str(doc)


# -------------------------------------------------------------- cross-encoder
@dataclass
class CrossEncoderReranker:
    """Sequence-classification cross-encoder on PyTorch-ROCm; for 1-label models the score
    is sigmoid(logit), as sentence_transformers.CrossEncoder.predict returns it; pairs are
    truncated to `max_length` and run in batches of `batch_size` (rerankers.py:93-116)."""
    model_name: str = "BAAI/bge-reranker-base"
    device: Optional[str] = None
    max_length: int = 512
    batch_size: int = 32
    fp16: bool = False  # sentence_transformers.CrossEncoder.predict runs fp32; half precision is an explicit knob

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
        model = AutoModelForSequenceClassification.from_pretrained(self.model_name, local_files_only=True)
        self._model = (model.half() if (self.fp16 and self._device.type == "cuda") else model).to(self._device).eval()

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        torch, out = self._torch, []
        with torch.inference_mode():
            for lo in range(0, len(docs), self.batch_size):
                part = docs[lo:lo + self.batch_size]
                enc = self._tok([query] * len(part), part, padding=True, truncation=True, max_length=self.max_length,
                                return_tensors="pt").to(self._device)
                logits = self._model(**enc).logits.float()
                vals = torch.sigmoid(logits[:, 0]) if logits.shape[-1] == 1 else logits[:, -1]
                out += [float(v) for v in vals.cpu()]
        return out

    def score(self, query: str, doc: str) -> float:
        return self.score_batch(query, [doc])[0]

    def score_pairs(self, pairs: Sequence[Tuple[str, str]]) -> List[float]:
        """(query, doc) pairs of SEVERAL queries through the model in full batches — the batched form behind
        HybridRetriever.search_batch (one forward per `batch_size` pairs instead of one short batch per query).  A
        forward's padding length and batch composition differ from the per-query score_batch call's: with a real
        cross-encoder the scores agree to floating-point noise of the model, not bit for bit (the stand-in scorer and
        the blend kernel are exact, so search_batch(qs)[i] == search(qs[i]) holds exactly in the tests)."""
        torch, out = self._torch, []
        with torch.inference_mode():
            for lo in range(0, len(pairs), self.batch_size):
                part = pairs[lo:lo + self.batch_size]
                enc = self._tok([q for q, _ in part], [d for _, d in part], padding=True, truncation=True,
                                max_length=self.max_length, return_tensors="pt").to(self._device)
                logits = self._model(**enc).logits.float()
                vals = torch.sigmoid(logits[:, 0]) if logits.shape[-1] == 1 else logits[:, -1]
                out += [float(v) for v in vals.cpu()]
        return out


# ------------------------------------------------------------------ LLM judges
LLM_RERANK_SYSTEM_PROMPT = (
    "You are a precise ranking model. Evaluate how well a candidate legal provision answers a user query. "
    'Output ONLY a JSON object {"score": float, "reason": "string"} with score between 0 and 1 '
    "(1.0 = directly answers the query, 0.0 = irrelevant); keep the reason short."
)
_NUMBER_0_1 = re.compile(r"([0-1](?:\.\d+)?)")


def build_llm_rerank_prompt(query: str, provision: str) -> str:
    return f"\nQuery:\n{query}\n\nCandidate provision:\n{provision}\n\nEvaluate relevance and return JSON only.\n"


def _judge_messages(query: str, doc: str) -> List[Dict[str, str]]:
    return [{"role": "system", "content": LLM_RERANK_SYSTEM_PROMPT},
            {"role": "user", "content": build_llm_rerank_prompt(query, doc)}]


def _shorten(s: str, n: int) -> str:
    return s if len(s) <= n else s[:n] + "…"


@dataclass
class LLMReranker:
    """Asks an LLM for {"score": x}; the reply is parsed as JSON first, else the first
    number in [0, 1] found in it, else 0 (rerankers.py:149-195)."""
    llm: Any
    temperature: float = 0.0
    max_query_chars: int = 800
    max_doc_chars: int = 2000

    @staticmethod
    def _extract_score(text: str) -> float:
        reply = (text or "").strip()
        try:
            parsed = json.loads(reply)
            if isinstance(parsed, dict) and "score" in parsed:
                return float(parsed["score"])
        except Exception:  # noqa: BLE001 - not JSON: fall through to the regex
            pass
        hit = _NUMBER_0_1.search(reply)
        return float(hit.group(1)) if hit else 0.0

    def _call_llm(self, query: str, doc: str) -> str:
        msgs = _judge_messages(_shorten(query, self.max_query_chars), _shorten(doc, self.max_doc_chars))
        return str(self.llm.chat(messages=msgs, tag="rerank_llm"))

    def score(self, query: str, doc: str) -> float:
        return _clip01(self._extract_score(self._call_llm(query, doc)))

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        return [self.score(query, d) for d in docs]


@dataclass
class CachedLLMReranker(LLMReranker):
    cache: Optional[Dict[Tuple[int, int], float]] = None

    def __post_init__(self):
        self.cache = {} if self.cache is None else self.cache

    def score(self, query: str, doc: str) -> float:
        key = (hash(query), hash(doc))
        if key not in self.cache:
            self.cache[key] = LLMReranker.score(self, query, doc)
        return self.cache[key]


@dataclass
class AsyncLLMReranker:
    llm: Any
    max_concurrency: int = 8
    base: Optional[LLMReranker] = None

    def __post_init__(self):
        self.base = self.base or LLMReranker(llm=self.llm)

    async def _call_llm_async(self, query: str, doc: str) -> str:
        if hasattr(self.llm, "achat"):
            return str(await self.llm.achat(messages=_judge_messages(query, doc), tag="rerank_llm"))
        loop = asyncio.get_running_loop()  # no async client: run the sync judge in the default executor
        return str(await loop.run_in_executor(None, self.base.score, query, doc))

    async def score(self, query: str, doc: str) -> float:
        return _clip01(LLMReranker._extract_score(await self._call_llm_async(query, doc)))

    async def score_batch(self, query: str, docs: List[str]) -> List[float]:
        gate = asyncio.Semaphore(self.max_concurrency)

        async def one(d: str) -> float:
            async with gate:
                return await self.score(query, d)
        return list(await asyncio.gather(*(one(d) for d in docs)))


@dataclass
class AsyncCachedLLMReranker(AsyncLLMReranker):
    cache: Optional[Dict[Tuple[int, int], float]] = None

    def __post_init__(self):
        AsyncLLMReranker.__post_init__(self)
        self.cache = {} if self.cache is None else self.cache

    async def score(self, query: str, doc: str) -> float:
        key = (hash(query), hash(doc))
        if key not in self.cache:
            self.cache[key] = await AsyncLLMReranker.score(self, query, doc)
        return self.cache[key]


# --------------------------------------------------------------------- factory
class RerankerFactory:
    """An LLM judge when an llm is given and top_k <= llm_threshold, otherwise the
    cross-encoder, loaded once per model name for the whole process (rerankers.py:281-312)."""
    _cross_cache: Dict[Any, Any] = {}

    def __init__(self, llm: Any = None, cross_model: str = "BAAI/bge-reranker-base", llm_threshold: int = 30,
                 use_cache: bool = True, device: Optional[str] = None, fp16: bool = False):
        self.llm, self.cross_model = llm, cross_model
        self.llm_threshold, self.use_cache = llm_threshold, use_cache
        self.device, self.fp16 = device, fp16
        self._cache: Dict[Tuple[int, int], float] = {}

    def create(self, top_k: int):
        if self.llm is not None and top_k <= self.llm_threshold:
            return CachedLLMReranker(llm=self.llm, cache=self._cache) if self.use_cache else LLMReranker(llm=self.llm)
        shared = type(self)._cross_cache
        key = (self.cross_model, self.device, self.fp16)
        if key not in shared:
            if self.cross_model == "hashing":  # explicit choice of the deterministic stand-in (encoders.py)
                from ..encoders import HashingCrossScorer
                shared[key] = HashingCrossScorer()
            else:
                shared[key] = CrossEncoderReranker(model_name=self.cross_model, device=self.device, fp16=self.fp16)
        return shared[key]


# ------------------------------------------------------------------ entry point
def rerank_candidates(query: str, candidates: Sequence[TextLike], reranker: BaseReranker, *, top_n: int,
                      content_key: str = "text", normalize: str = "minmax", sigmoid_temperature: float = 1.0,
                      include_debug: bool = False) -> List[Tuple[TextLike, RerankResult]]:
    """Score every candidate, normalise ("minmax" default, "sigmoid", "none"), return the
    top_n (candidate, RerankResult) pairs by normalised score, stable (rerankers.py:319-350)."""
    if top_n <= 0:
        return []
    raw = reranker.score_batch(query, [_to_doc_text(c, content_key=content_key) for c in candidates])
    if normalize == "sigmoid":
        norm = sigmoid_calibrate(raw, temperature=sigmoid_temperature)
    elif normalize == "none":
        norm = list(raw)
    else:
        norm = minmax_normalize(raw)
    scored = [(cand, RerankResult(raw_score=r, norm_score=v, meta={"raw": r, "norm": v} if include_debug else {}))
              for cand, r, v in zip(candidates, raw, norm)]
    scored.sort(key=lambda pair: pair[1].norm_score, reverse=True)
    return scored[:top_n]
