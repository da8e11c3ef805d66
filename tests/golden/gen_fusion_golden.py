#!/usr/bin/env python3
"""Generate golden vectors for fusion / normalisation / rerank-blend / dedup /
the whole HybridRetriever.search() orchestration by running the REFERENCE'S
OWN CODE on synthetic channel outputs.

Runs ONLY in the build container (reference mounted read-only at
/root/reference).  The reference's retrieval modules import third-party wheels
that are absent here (jieba, rank_bm25, faiss, FlagEmbedding, colbert, openai);
none of the functions exercised below touch them, so empty placeholder modules
are registered in sys.modules purely to let the `import` statements succeed
(SURVEY.md §8c records this as verified).  Only the generated vectors (inputs
+ outputs, JSON) are committed; no reference source is copied.

    PYTHONHASHSEED=0 PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/gen_fusion_golden.py

Exercised reference symbols (file:line in /root/reference/legalrag/retrieval):
  hybrid_retriever.py:24   _minmax
  hybrid_retriever.py:33   _rrf_with_breakdown
  hybrid_retriever.py:71   _dedup_keep_best
  hybrid_retriever.py:181-245 search_dense / search_bm25 / search_colbert
  hybrid_retriever.py:282  HybridRetriever.search
  hybrid_retriever.py:389  HybridRetriever._fuse
  rerankers.py:48-67       minmax_normalize / sigmoid / sigmoid_calibrate
  rerankers.py:319         rerank_candidates
"""
from __future__ import annotations

import json
import re
import sys
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

class _Placeholder(types.ModuleType):
    """Absent third-party wheel: any attribute resolves to a dummy class."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (Exception,), {})


for name in ("jieba", "rank_bm25", "faiss", "FlagEmbedding", "colbert", "colbert.infra", "openai"):
    sys.modules.setdefault(name, _Placeholder(name))

from legalrag.config import AppConfig  # noqa: E402
from legalrag.schemas import LawChunk, RetrievalHit  # noqa: E402
from legalrag.retrieval import hybrid_retriever as hr  # noqa: E402
from legalrag.retrieval import rerankers as rr  # noqa: E402

OUT = Path(__file__).resolve().parent


def mk_chunk(i: int) -> LawChunk:
    return LawChunk(
        id=f"src.txt::{i}",
        law_name="Synthetic Code",
        article_no=f"§ {i}",
        article_id=str(i),
        text=f"text of provision {i}",
        lang="en",
        source="src.txt",
    )


def channel(rng, pool, n, lo, hi, *, equal=False, f32=False):
    """n distinct ids from pool with distinct descending-ish random scores."""
    if n == 0:
        return []
    ids = rng.choice(pool, size=n, replace=False).tolist()
    if equal:
        sc = [float(lo)] * n
    else:
        sc = rng.uniform(lo, hi, size=n)
        if f32:
            sc = sc.astype(np.float32)
        sc = [float(x) for x in sc]
    return [[int(i), s] for i, s in zip(ids, sc)]


def hits_from(pairs, chunks, ch):
    out = []
    for r, (i, s) in enumerate(pairs, start=1):
        out.append(
            RetrievalHit(
                chunk=chunks[i], score=float(s), rank=r, source="retriever",
                score_breakdown={"channel": [ch], f"{ch}_raw": float(s)},
            )
        )
    return out


def dump_hit(h: RetrievalHit):
    return {
        "id": h.chunk.id,
        "score": float(h.score),
        "rank": h.rank,
        "source": h.source,
        "breakdown": h.score_breakdown,
    }


def new_retriever(**knobs):
    cfg = AppConfig()
    for k, v in knobs.items():
        setattr(cfg.retrieval, k, v)
    r = hr.HybridRetriever.__new__(hr.HybridRetriever)
    r.cfg = cfg
    r.dense = None
    r.bm25 = None
    r.colbert = None
    r.graph = None
    return r


def gen_fuse_cases():
    rng = np.random.default_rng(20260304)
    chunks = {i: mk_chunk(i) for i in range(400)}
    pool = np.arange(400)
    cases = []
    shapes = [
        ("3ch_k10", 10, 10, 10, {}),
        ("3ch_k10_overlap", 10, 10, 10, {"small_pool": 18}),
        ("2ch_k10_no_colbert", 10, 10, 0, {}),
        ("dense_only_k10", 10, 0, 0, {}),
        ("bm25_only_k7", 0, 7, 0, {}),
        ("empty_bm25", 10, 0, 10, {}),
        ("all_empty", 0, 0, 0, {}),
        ("ragged_k3_k10_k5", 3, 10, 5, {"small_pool": 14}),
        ("3ch_k80", 80, 80, 80, {"small_pool": 160}),
        ("bm25_all_equal", 10, 10, 10, {"bm25_equal": True, "small_pool": 20}),
        ("single_hit_each", 1, 1, 1, {"small_pool": 2}),
    ]
    methods = ["rrf_norm_blend", "rrf", "wrrf", "weighted_sum"]
    for name, nd, nb, nc, opt in shapes:
        p = pool[: opt["small_pool"]] if "small_pool" in opt else pool
        d = channel(rng, p, nd, -0.2, 0.9, f32=True)
        b = channel(rng, p, nb, 0.0, 40.0, equal=opt.get("bm25_equal", False))
        c = channel(rng, p, nc, 5.0, 30.0, f32=True)
        for m in methods:
            knobs = {"fusion_method": m}
            if name == "3ch_k10_overlap" and m == "rrf_norm_blend":
                knobs.update({"rrf_alpha": 0.3, "rrf_k": 10, "dense_weight": 0.5,
                              "bm25_weight": 0.2, "colbert_weight": 0.9})
            r = new_retriever(**knobs)
            out = r._fuse(
                dense_hits=hits_from(d, chunks, "dense"),
                bm25_hits=hits_from(b, chunks, "bm25"),
                colbert_hits=hits_from(c, chunks, "colbert"),
            )
            cases.append({
                "name": f"{name}__{m}",
                "knobs": knobs,
                "dense": d, "bm25": b, "colbert": c,
                "expected": [dump_hit(h) for h in out],
            })
    return cases


class FakeDense:
    def __init__(self, pairs, chunks):
        self.pairs, self.chunks = pairs, chunks

    def search(self, query, top_k):
        # mirrors DenseRetriever.search output shape (dense_retriever.py:46-59)
        out = []
        for rank, (i, s) in enumerate(self.pairs[: int(top_k)], start=1):
            out.append(RetrievalHit(chunk=self.chunks[i], score=float(s), rank=rank,
                                    source="retriever", semantic_score=float(s)))
        return out


class FakePairs:
    def __init__(self, pairs, chunks):
        self.pairs, self.chunks = pairs, chunks

    def search(self, query, top_k):
        return [(self.chunks[i], float(s)) for i, s in self.pairs[: int(top_k)]]


class FakeReranker:
    """Deterministic stand-in for the cross-encoder.  NOTE (reference quirk):
    HybridRetriever.search passes RetrievalHit objects as `candidates`
    (hybrid_retriever.py:343) and rerankers._to_doc_text falls through to
    str(doc) for them (rerankers.py:78-86), so the text the cross-encoder sees
    is the pydantic repr of the whole hit, not chunk.text.  The table is keyed
    by whatever key can be recovered from that string, and the exact strings
    are recorded so the restatement can be checked to pass identical ones."""

    def __init__(self, table):
        self.table = table
        self.seen_docs = []

    def _key(self, doc):
        if doc in self.table:
            return doc
        m = re.search(r"LawChunk\(id='([^']+)'", doc)
        return m.group(1) if m else doc

    def score(self, query, doc):
        return float(self.table[self._key(doc)])

    def score_batch(self, query, docs):
        self.seen_docs.extend(docs)
        return [float(self.table[self._key(d)]) for d in docs]


def sort_desc(pairs):
    return sorted(pairs, key=lambda p: -p[1])


def gen_search_cases():
    rng = np.random.default_rng(777)
    chunks = {i: mk_chunk(i) for i in range(300)}
    cases = []
    specs = [
        ("default_topk10", dict(), 10, 30, (10, 10, 10)),
        ("topk5_cfg10", dict(), 5, 30, (10, 10, 10)),
        ("topk20_gt_cfg", dict(), 20, 60, (20, 20, 20)),
        ("no_rerank", dict(enable_rerank=False), 10, 30, (10, 10, 10)),
        ("no_colbert_rerank", dict(), 10, 24, (10, 10, 0)),
        ("min_final_0", dict(min_final_score=0.0), 10, 30, (10, 10, 10)),
        ("min_final_high", dict(min_final_score=0.6), 10, 30, (10, 10, 10)),
        ("beta_07_topn5", dict(rerank_beta=0.7, rerank_top_n=5), 10, 30, (10, 10, 10)),
        ("weighted_sum", dict(fusion_method="weighted_sum"), 10, 30, (10, 10, 10)),
        ("rrf", dict(fusion_method="rrf", min_final_score=0.05), 10, 30, (10, 10, 10)),
        ("dense_only", dict(), 10, 40, (10, 0, 0)),
        ("all_filtered", dict(min_final_score=5.0), 10, 30, (10, 10, 10)),
    ]
    for name, knobs, top_k, pool_n, (nd, nb, nc) in specs:
        pool = np.arange(pool_n)
        d = sort_desc(channel(rng, pool, nd, 0.1, 0.9, f32=True))
        b = sort_desc(channel(rng, pool, nb, 0.0, 40.0))
        c = sort_desc(channel(rng, pool, nc, 5.0, 30.0, f32=True))
        ce = {chunks[i].id: float(x) for i, x in zip(range(pool_n), rng.uniform(0.0, 1.0, size=pool_n))}
        r = new_retriever(enable_graph=False, **knobs)
        r.dense = FakeDense(d, chunks)
        r.bm25 = FakePairs(b, chunks)
        r.colbert = FakePairs(c, chunks) if nc else None
        fake = FakeReranker(ce)
        orig = hr.RerankerFactory.create
        hr.RerankerFactory.create = lambda self, top_k, _f=fake: _f
        try:
            out = r.search("synthetic question", llm=None, top_k=top_k, decision=None)
        finally:
            hr.RerankerFactory.create = orig
        cases.append({
            "name": name, "knobs": dict(enable_graph=False, **knobs), "top_k": top_k,
            "dense": d, "bm25": b, "colbert": c,
            "ce_raw_by_id": ce,
            "ce_docs_seen": list(fake.seen_docs),
            "expected": [dump_hit(h) for h in out],
        })
    return cases


def gen_util_cases():
    rng = np.random.default_rng(5)
    out = {"minmax": [], "rrf": [], "dedup": [], "rerank_norm": [], "rerank_candidates": []}
    vecs = [[], [3.0], [2.0, 2.0, 2.0], [1.0, 1.0 + 1e-13], [0.1, 0.7, 0.3],
            rng.normal(size=17).tolist(), rng.uniform(0, 40, size=80).tolist(),
            [float(np.float32(x)) for x in rng.uniform(-1, 1, size=10)]]
    for v in vecs:
        out["minmax"].append({"in": v, "hybrid_minmax": hr._minmax(v),
                              "rerank_minmax": rr.minmax_normalize(v)})
    xs = [-800.0, -30.5, -1.0, -1e-9, 0.0, 1e-9, 0.3, 2.0, 30.5, 800.0]
    out["rerank_norm"] = {
        "x": xs,
        "sigmoid": [rr.sigmoid(x) for x in xs],
        "calibrate_t1": rr.sigmoid_calibrate(xs, 1.0),
        "calibrate_t0p25": rr.sigmoid_calibrate(xs, 0.25),
        "calibrate_t0": rr.sigmoid_calibrate(xs, 0.0),
    }
    lists = {"dense": ["a", "b", "c", "d"], "bm25": ["c", "a", "e"], "colbert": ["f", "a"]}
    for k, w in [(60, None), (10, {"dense": 0.6, "bm25": 0.4, "colbert": 0.35})]:
        tot, con = hr._rrf_with_breakdown(lists, k=k, weights=w)
        out["rrf"].append({"lists": lists, "k": k, "weights": w, "totals": tot, "contrib": con})
    # rerank_candidates (rerankers.py:319) with dict candidates + each normaliser
    cands = [{"text": f"doc {i}", "id": i} for i in range(9)]
    table = {f"doc {i}": float(s) for i, s in enumerate(rng.normal(size=9))}
    for norm in ("minmax", "sigmoid", "none"):
        for top_n in (9, 4, 0):
            res = rr.rerank_candidates("q", cands, FakeReranker(table), top_n=top_n,
                                       normalize=norm, sigmoid_temperature=0.5, include_debug=True)
            out["rerank_candidates"].append({
                "normalize": norm, "top_n": top_n, "raw_by_text": table,
                "expected": [{"id": c["id"], "raw": r_.raw_score, "norm": r_.norm_score, "meta": r_.meta}
                             for c, r_ in res],
            })
    # dedup: duplicates with channel/contrib unions (hybrid_retriever.py:71-130)
    chunks = {i: mk_chunk(i) for i in range(6)}
    hits = [
        RetrievalHit(chunk=chunks[0], score=0.9, rank=1, score_breakdown={"channel": ["dense"], "channel_contrib": {"dense": 0.5}}),
        RetrievalHit(chunk=chunks[1], score=0.8, rank=2, score_breakdown={"channel": "bm25"}),
        RetrievalHit(chunk=chunks[0], score=0.95, rank=3, score_breakdown={"channel": ["bm25"], "channel_contrib": {"bm25": 0.7, "dense": 0.1}}),
        RetrievalHit(chunk=chunks[2], score=0.1, rank=4, score_breakdown=None),
        RetrievalHit(chunk=chunks[1], score=0.3, rank=5, score_breakdown={"channel": ["colbert", "dense"]}),
        RetrievalHit(chunk=chunks[3], score=0.85, rank=6, score_breakdown={"channel": ("graph",)}),
    ]
    inp = [dump_hit(h) for h in hits]
    inp = json.loads(json.dumps(inp))
    res = hr._dedup_keep_best(hits)
    out["dedup"].append({"in": inp, "expected": [dump_hit(h) for h in res]})
    return out


def main():
    fuse = gen_fuse_cases()
    search = gen_search_cases()
    util = gen_util_cases()
    (OUT / "fusion_golden.json").write_text(json.dumps({"cases": fuse}, indent=0, ensure_ascii=False))
    (OUT / "search_golden.json").write_text(json.dumps({"cases": search}, indent=0, ensure_ascii=False))
    (OUT / "util_golden.json").write_text(json.dumps(util, indent=0, ensure_ascii=False))
    print("fuse cases", len(fuse), "search cases", len(search))
    for c in search:
        print(c["name"], [(h["id"].split("::")[1], round(h["score"], 4), h["source"]) for h in c["expected"][:4]])


if __name__ == "__main__":
    main()
