"""Retrieval metrics + a deterministic offline query set.

Metrics restate scripts/evaluate_retrieval.py:30-62 (hit/recall/mrr/ndcg@k and
the unique-preserving article_id extraction); the committed script forgets
`import math` (:49) — fixed here.  The reference's evaluation queries come from
an LLM-driven generator (scripts/generate_synthetic_data.py) that cannot run
offline, so `synthetic_queries` builds a seeded stand-in set from the corpus
itself: for every section a TITLE query (the heading after "§ x-yyy.") and a
SPAN query (a seeded 12-word window of the body); gold = that section's
article_id.  Chinese articles (no word separators) get a seeded 16-character
window of the article body instead.
"""
from __future__ import annotations

import math
import re
from typing import Any, Dict, List, Sequence, Set, Tuple

import numpy as np


def hit_at_k(pred: List[str], gold: Set[str], k: int) -> float:
    return float(any(h.strip() in gold for h in pred[:k]))


def recall_at_k(pred: List[str], gold: Set[str], k: int) -> float:
    if not gold:
        return 0.0
    return len(set(pred[:k]) & gold) / len(gold)


def mrr_at_k(pred: List[str], gold: Set[str], k: int) -> float:
    for i, x in enumerate(pred[:k], 1):
        if x in gold:
            return 1.0 / i
    return 0.0


def ndcg_at_k(pred: List[str], gold: Set[str], k: int) -> float:
    def dcg(xs: Sequence[str]) -> float:
        return sum((1.0 if x in gold else 0.0) / math.log2(i + 1) for i, x in enumerate(xs[:k], 1))
    ideal = dcg(list(gold))
    if ideal <= 1e-12:
        return 0.0
    return dcg(pred) / ideal


def get_hit_ids(hits: List[Any]) -> List[str]:
    return list(dict.fromkeys(str(getattr(h.chunk, "article_id", "") or "")
                              for h in hits if getattr(h.chunk, "article_id", "")))


def all_metrics(pred: List[str], gold: Set[str]) -> Dict[str, float]:
    return {"R@5": recall_at_k(pred, gold, 5), "R@10": recall_at_k(pred, gold, 10),
            "MRR@10": mrr_at_k(pred, gold, 10), "nDCG@10": ndcg_at_k(pred, gold, 10),
            "Hit@3": hit_at_k(pred, gold, 3), "Hit@10": hit_at_k(pred, gold, 10)}


_TITLE_RE = re.compile(r"^§\s*[\w\-\.]+?\.\s+(.+?)\.(?:\s|$)")


def synthetic_queries(chunks, seed: int = 0, span_words: int = 12, span_chars: int = 16) -> List[Tuple[str, str, str]]:
    """[(query, gold article_id, kind)] — deterministic for a given corpus + seed."""
    rng = np.random.default_rng(seed)
    out: List[Tuple[str, str, str]] = []
    for c in chunks:
        if (getattr(c, "lang", None) or "").lower() == "zh":
            body = c.text.split(" ", 1)[-1].strip()  # drop the "第N条" head
            if len(body) >= span_chars + 6:
                s = int(rng.integers(0, len(body) - span_chars))
                out.append((body[s:s + span_chars], c.article_id, "span_zh"))
            continue
        m = _TITLE_RE.match(c.text)
        if m and 3 <= len(m.group(1)) <= 200:
            out.append((m.group(1).strip(), c.article_id, "title"))
        words = c.text.split()
        if len(words) >= span_words + 8:
            s = int(rng.integers(6, len(words) - span_words))
            out.append((" ".join(words[s:s + span_words]), c.article_id, "span"))
    return out
