"""Fusion / normalisation / rerank blend / dedup / search() orchestration.

Plain-Python (float64) restatement of
  legalrag/retrieval/hybrid_retriever.py:24-30   _minmax
  legalrag/retrieval/hybrid_retriever.py:33-56   _rrf_with_breakdown
  legalrag/retrieval/hybrid_retriever.py:71-130  _dedup_keep_best
  legalrag/retrieval/hybrid_retriever.py:181-245 per-channel wrappers
  legalrag/retrieval/hybrid_retriever.py:282-384 HybridRetriever.search
  legalrag/retrieval/hybrid_retriever.py:389-551 HybridRetriever._fuse
  legalrag/retrieval/rerankers.py:48-67,319-350  normalisers, rerank_candidates
PINNED: bit-exact against tests/golden/{fusion,search,util}_golden.json, which
were produced by running the reference's own functions.

A hit is a dict {"id", "score", "rank", "source", "breakdown"}.
Deliberate determinisation: the reference iterates Python `set`s of ids
(hybrid_retriever.py:460,484,526), so the order of EXACTLY tied fused scores
depends on PYTHONHASHSEED; here ties keep first-appearance order
(dense list, then bm25, then colbert).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

CHANNELS = ("dense", "bm25", "colbert")

DEFAULTS = dict(
    top_k=10, dense_weight=0.6, bm25_weight=0.4, colbert_weight=0.35, min_final_score=0.2,
    enable_rerank=True, rerank_top_n=30, rrf_alpha=0.5, rerank_beta=0.35,
    fusion_method="rrf_norm_blend", rrf_k=60,
)


def minmax(scores: Sequence[float]) -> List[float]:
    if not scores:
        return []
    lo, hi = min(scores), max(scores)
    if hi - lo < 1e-12:
        return [0.0 for _ in scores]
    return [(float(s) - lo) / (hi - lo) for s in scores]


def sigmoid(x: float) -> float:
    if x >= 0:
        z = math.exp(-x)
        return 1.0 / (1.0 + z)
    z = math.exp(x)
    return z / (1.0 + z)


def sigmoid_calibrate(scores: Sequence[float], temperature: float = 1.0) -> List[float]:
    t = max(1e-6, float(temperature))
    return [sigmoid(s / t) for s in scores]


def rrf_with_breakdown(rank_lists: Dict[str, List[str]], *, k: int = 60, weights=None):
    totals: Dict[str, float] = {}
    contrib: Dict[str, Dict[str, float]] = {}
    weights = weights or {}
    for channel, ids in rank_lists.items():
        w = float(weights.get(channel, 1.0))
        for rank, cid in enumerate(ids, start=1):
            v = w * (1.0 / (k + rank))
            totals[cid] = totals.get(cid, 0.0) + v
            contrib.setdefault(cid, {})
            contrib[cid][channel] = v
    return totals, contrib


def _stable_desc(pairs):
    return sorted(pairs, key=lambda p: float(p[1]), reverse=True)


def channel_hits(pairs: Sequence[Tuple[str, float]], channel: str) -> List[dict]:
    """search_dense/search_bm25/search_colbert wrappers (hybrid_retriever.py:181-245):
    stable re-sort by score desc, ranks 1.., breakdown {"channel":[ch], "<ch>_raw": s}."""
    hits = [{"id": i, "score": float(s), "rank": 0, "source": "retriever",
             "breakdown": {"channel": [channel], f"{channel}_raw": float(s)}} for i, s in pairs]
    hits.sort(key=lambda h: h["score"], reverse=True)
    for r, h in enumerate(hits, start=1):
        h["rank"] = r
    return hits


def fuse(dense: Sequence[Tuple[str, float]], bm25: Sequence[Tuple[str, float]],
         colbert: Sequence[Tuple[str, float]], knobs: Optional[dict] = None) -> List[dict]:
    kn = dict(DEFAULTS)
    kn.update(knobs or {})
    method = str(kn["fusion_method"]).lower()
    rrf_k = int(kn["rrf_k"])
    alpha = float(kn["rrf_alpha"])
    weights = {"dense": float(kn["dense_weight"]), "bm25": float(kn["bm25_weight"]),
               "colbert": float(kn["colbert_weight"])}

    lists = {"dense": _stable_desc(dense), "bm25": _stable_desc(bm25), "colbert": _stable_desc(colbert)}
    rank_lists = {ch: [i for i, _ in lists[ch]] for ch in CHANNELS}

    channels_by_id: Dict[str, List[str]] = {}
    for ch in CHANNELS:
        for cid in rank_lists[ch]:
            channels_by_id.setdefault(cid, [])
            if ch not in channels_by_id[cid]:
                channels_by_id[cid].append(ch)

    norm_map: Dict[str, Dict[str, float]] = {}
    for ch in CHANNELS:
        vals = minmax([float(s) for _, s in lists[ch]])
        m: Dict[str, float] = {}
        for i, (cid, _) in enumerate(lists[ch]):
            m[cid] = float(vals[i])  # later duplicate of an id overwrites, as the dict does
        norm_map[ch] = m

    if method == "wrrf":
        rrf_total, rrf_raw = rrf_with_breakdown(rank_lists, k=rrf_k, weights=weights)
    else:
        rrf_total, rrf_raw = rrf_with_breakdown(rank_lists, k=rrf_k)

    rrf_norm_map: Dict[str, float] = {}
    if rrf_total:
        items = list(rrf_total.items())
        vals = minmax([float(v) for _, v in items])
        for i, (cid, _) in enumerate(items):
            rrf_norm_map[cid] = float(vals[i])

    all_ids: List[str] = list(rrf_total.keys())  # first-appearance order (see module doc)

    def rrf_alloc(cid, mass):
        raw = rrf_raw.get(cid, {}) or {}
        total = float(rrf_total.get(cid, 0.0))
        if mass <= 0.0 or total <= 1e-18:
            return {}
        return {str(ch): mass * float(v) / total for ch, v in raw.items()}

    rows = []
    for cid in all_ids:
        norms = {ch: float(norm_map[ch].get(cid, 0.0)) for ch in weights}
        w_terms = {ch: float(weights[ch]) * float(norms[ch]) for ch in weights}
        wsum = sum(w_terms.values())
        rrf_norm = float(rrf_norm_map.get(cid, 0.0))
        contrib = {ch: 0.0 for ch in weights}
        if method == "weighted_sum":
            score = float(wsum)
            contrib.update(w_terms)
        elif method in ("rrf", "wrrf"):
            score = float(rrf_norm)
            contrib.update(rrf_alloc(cid, score))
        else:
            score = float(alpha) * float(rrf_norm) + (1.0 - float(alpha)) * float(wsum)
            for ch, v in w_terms.items():
                contrib[ch] += (1.0 - float(alpha)) * float(v)
            for ch, v in rrf_alloc(cid, float(alpha) * float(rrf_norm)).items():
                contrib[ch] = contrib.get(ch, 0.0) + float(v)
        membership = list(channels_by_id.get(cid, []))
        ch_list = sorted(membership, key=lambda c: (float(contrib.get(str(c), 0.0)), str(c)), reverse=True)
        rows.append((cid, float(score), contrib, ch_list, rrf_norm, float(wsum), norms))

    rows.sort(key=lambda r: r[1], reverse=True)
    out = []
    for r, (cid, score, contrib, ch_list, rrf_norm, wsum, norms) in enumerate(rows, start=1):
        sb = {
            "fusion_method": method, "rrf_k": int(rrf_k), "alpha": float(alpha),
            "channel_weights": dict(weights), "channel": ch_list, "channel_contrib": contrib,
            "rrf_norm": rrf_norm, "weighted_sum": wsum,
            "dense_norm": norms["dense"], "bm25_norm": norms["bm25"], "colbert_norm": norms["colbert"],
        }
        out.append({"id": cid, "score": score, "rank": r, "source": "retriever", "breakdown": sb})
    return out


def _as_channel_list(x):
    if x is None:
        return []
    if isinstance(x, (list, set, tuple)):
        return [str(i) for i in x]
    return [str(x)]


def dedup_keep_best(hits: List[dict]) -> List[dict]:
    best: Dict[str, dict] = {}
    for h in hits:
        cid = h["id"]
        sb = h["breakdown"] or {}
        if cid not in best:
            if "channel" in sb:
                sb["channel"] = _as_channel_list(sb.get("channel"))
                h["breakdown"] = sb
            best[cid] = h
            continue
        b = best[cid]
        sb_best = b["breakdown"] or {}
        chs: List[str] = []
        for c in _as_channel_list(sb_best.get("channel")) + _as_channel_list(sb.get("channel")):
            if c not in chs:
                chs.append(c)
        merged_contrib: Dict[str, float] = {}
        for src in (sb_best.get("channel_contrib", {}) or {}, sb.get("channel_contrib", {}) or {}):
            for k, v in src.items():
                merged_contrib[str(k)] = merged_contrib.get(str(k), 0.0) + float(v)
        if float(h["score"]) > float(b["score"]):
            best[cid] = h
        rep = best[cid]
        sb_rep = rep["breakdown"] or {}
        if merged_contrib:
            chs.sort(key=lambda c: float(merged_contrib.get(c, 0.0)), reverse=True)
            sb_rep["channel_contrib"] = merged_contrib
        else:
            chs.sort()
        sb_rep["channel"] = chs
        rep["breakdown"] = sb_rep
    out = list(best.values())
    out.sort(key=lambda x: float(x["score"]), reverse=True)
    for i, h in enumerate(out, start=1):
        h["rank"] = i
    return out


def rerank_blend(fused: List[dict], raw_scores: Sequence[float], beta: float) -> List[dict]:
    """hybrid_retriever.py:338-355 given CE raw scores for fused[:len(raw_scores)]."""
    n = len(raw_scores)
    cand = fused[:n]
    norm = minmax(list(raw_scores))
    results = list(zip(cand, raw_scores, norm))
    results.sort(key=lambda x: x[2], reverse=True)  # rerankers.py:349 (stable)
    new_hits = []
    for hit, rs, ns in results:
        hit["breakdown"] = hit["breakdown"] or {}
        hit["breakdown"].update({"rerank_raw": rs, "rerank_norm": ns, "rerank_beta": beta})
        hit["score"] = (1 - beta) * float(hit["score"]) + beta * float(ns)
        hit["source"] = "rerank"
        new_hits.append(hit)
    fused[: len(new_hits)] = new_hits
    fused.sort(key=lambda x: float(x["score"]), reverse=True)
    for i, h in enumerate(fused, start=1):
        h["rank"] = i
    return fused


def search(dense: Sequence[Tuple[str, float]], bm25: Sequence[Tuple[str, float]],
           colbert: Sequence[Tuple[str, float]], *, top_k: int = 10, knobs: Optional[dict] = None,
           ce_score: Optional[Callable[[List[str]], List[float]]] = None) -> List[dict]:
    """HybridRetriever.search (hybrid_retriever.py:282-384) downstream of the
    channels (graph branch excluded: out of scope).  `dense` etc.
    are the channel outputs ALREADY cut to eff_top_k; `ce_score(ids)` returns
    the cross-encoder raw score for each candidate id."""
    kn = dict(DEFAULTS)
    kn.update(knobs or {})
    top_k = max(1, int(top_k))
    fused = fuse(dense, bm25, colbert, kn)
    min_final = float(kn["min_final_score"])
    fused = [h for h in fused if float(h["score"]) >= min_final]
    if kn["enable_rerank"]:
        n = int(kn["rerank_top_n"])
        cand = fused[:n]
        if cand:
            raw = ce_score([h["id"] for h in cand])
            fused = rerank_blend(fused, raw, float(kn["rerank_beta"]))
    fused = dedup_keep_best(fused)
    return fused[:top_k]


def eff_top_k(top_k: int, cfg_top_k: Optional[int]) -> int:
    """hybrid_retriever.py:284-291."""
    top_k = max(1, int(top_k))
    eff = int(cfg_top_k or (top_k * 8)) if cfg_top_k is not None else top_k * 8
    if eff < top_k:
        eff = top_k
    return eff
