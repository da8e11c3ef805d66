"""The `/retrieve` contract of the retrieval micro-service as plain host functions
(legalrag/services/retrieval_api.py:51-77; the same clamp sits in RagPipeline.retrieve,
rag_pipeline.py:249-251).  Routing, the LLM gateway and the FastAPI app itself stay the
reference's: a deployment keeps its own `@app.post("/retrieve")` and calls `retrieve_response`
with its retriever, router and config, or binds it with `make_route`.  Errors are raised as
`ServiceError(status_code, detail)` with the reference's codes and messages, for the app layer to
turn into HTTPException."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional


class ServiceError(Exception):
    def __init__(self, status_code: int, detail: str):
        super().__init__(detail)
        self.status_code, self.detail = int(status_code), str(detail)


def effective_top_k(top_k: Any, top_k_factor: float = 1.0) -> int:
    """clamp(int(top_k * factor), 3, 30) — retrieval_api.py:68-69, rag_pipeline.py:249-251."""
    eff = int(top_k * top_k_factor)
    return max(3, min(eff, 30))


def _dump(x: Any) -> Any:
    return x.model_dump() if hasattr(x, "model_dump") else x


def serialize_hits(hits: List[Any]) -> List[Dict[str, Any]]:
    """retrieval_api.py:24-28."""
    return [_dump(h) for h in hits]


def retrieve_response(body: Dict[str, Any], retriever: Any, router: Any, cfg: Any) -> Dict[str, Any]:
    """Body of `POST /retrieve`: {"question", "top_k"?} -> {"question", "top_k", "decision", "hits"}."""
    if retriever is None or router is None or cfg is None:
        raise ServiceError(503, "retriever not ready")
    question = (body.get("question") or "").strip()
    if not question:
        raise ServiceError(400, "Missing 'question'")
    top_k = body.get("top_k")
    try:
        top_k = int(top_k) if top_k is not None else cfg.retrieval.top_k
    except Exception:  # noqa: BLE001 - any unparsable value falls back to the configured depth (:62-65)
        top_k = cfg.retrieval.top_k
    decision = router.route(question)
    eff_top_k = effective_top_k(top_k, getattr(decision, "top_k_factor", 1.0))
    hits = retriever.search(question, top_k=eff_top_k, decision=decision)
    return {"question": question, "top_k": eff_top_k, "decision": _dump(decision), "hits": serialize_hits(hits)}


def make_route(retriever: Any, router: Any, cfg: Any, http_exception: Optional[Callable[..., Exception]] = None):
    """A route function for `app.post("/retrieve")(...)`; `http_exception` = fastapi.HTTPException."""
    def route(body: Dict[str, Any]):
        try:
            return retrieve_response(body, retriever, router, cfg)
        except ServiceError as e:
            if http_exception is not None:
                raise http_exception(status_code=e.status_code, detail=e.detail) from None
            raise
    return route
