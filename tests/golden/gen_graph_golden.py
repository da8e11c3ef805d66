#!/usr/bin/env python3
"""Generate golden vectors for the graph channel by running the REFERENCE'S OWN
CODE (read-only mount /root/reference) on a synthetic law graph.

    PYTHONHASHSEED=0 PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/gen_graph_golden.py

Exercised reference symbols:
  legalrag/retrieval/graph_store.py:21-87    LawGraphStore.load
  legalrag/retrieval/graph_store.py:89-169   LawGraphStore.walk
  legalrag/retrieval/graph_store.py:171-196  LawGraphStore.get_neighbors
  legalrag/retrieval/graph_retriever.py:19-46  _cosine_sim / _depth_decay / _relation_weight
  legalrag/retrieval/graph_retriever.py:82-219 GraphRetriever.search  (store replaced by a
      stand-in whose `_embed` returns seeded vectors: the reference's FlagModel/faiss wheels
      are absent; every line of the walk -> hydrate -> score -> sort path is the reference's)
  legalrag/retrieval/hybrid_retriever.py:247-384 HybridRetriever.search_graph / .search with a
      GRAPH_AUGMENTED routing decision (channels and cross-encoder faked as in
      gen_fusion_golden.py) -> "hybrid" section

Absent third-party wheels are registered as empty placeholder modules so that the
`import` statements succeed (same device as gen_fusion_golden.py).  Only the synthetic
graph (graph/law_graph_fixture.jsonl — data made HERE) and the inputs + outputs
(graph_golden.json) are committed.
"""
from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")


class _Placeholder(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (Exception,), {})


for name in ("jieba", "rank_bm25", "faiss", "FlagEmbedding", "colbert", "colbert.infra", "openai"):
    sys.modules.setdefault(name, _Placeholder(name))

from legalrag.config import AppConfig  # noqa: E402
from legalrag.schemas import LawChunk, RetrievalHit  # noqa: E402
from legalrag.retrieval.graph_store import LawGraphStore  # noqa: E402
from legalrag.retrieval import graph_retriever as gr  # noqa: E402
from legalrag.retrieval import hybrid_retriever as hr  # noqa: E402

OUT = Path(__file__).resolve().parent
GRAPH = OUT / "graph" / "law_graph_fixture.jsonl"
DIM = 48
N = 60
RELS = ["defined_by", "defines_term", "cite", "cited_by", "prev", "next", "neighbor", "amend", "ref"]


def build_graph():
    """60 articles; prev/next chain, seeded cross references, a few string-form neighbours,
    dangling targets, duplicate edges, and edges with / without evidence."""
    rng = np.random.default_rng(424242)
    rows = []
    for i in range(N):
        nbs = []
        if i + 1 < N:
            nbs.append({"article_id": str(i + 1), "relation": "next", "conf": 1.0})
        if i > 0:
            nbs.append({"article_id": str(i - 1), "relation": "prev", "conf": 1.0})
        for _ in range(int(rng.integers(0, 4))):
            j = int(rng.integers(0, N + 3))  # ids >= N do not exist
            rel = RELS[int(rng.integers(0, len(RELS)))]
            e = {"article_id": str(j), "relation": rel, "conf": round(float(rng.uniform(0.2, 1.0)), 3)}
            if rng.random() < 0.6:
                e["evidence"] = {"span": [int(rng.integers(0, 50)), int(rng.integers(50, 99))]}
            nbs.append(e)
        if i % 11 == 0:
            nbs.append(str((i * 7 + 3) % N))  # bare-string neighbour form
        if i % 13 == 0:
            nbs.append({"id": str((i * 5 + 1) % N), "relation": "cite"})  # "id" key, default conf
        rows.append({"article_id": str(i), "article_no": f"§ {i}", "law_name": "Synthetic Code",
                     "title": f"Title {i}", "chapter": f"Ch {i // 10}", "section": None, "neighbors": nbs,
                     "meta": {"k": i}})
    rows.append({"id": "X1", "article_no": "", "neighbors": [{"article_id": "3", "relation": "cite", "conf": 0.9,
                                                              "evidence": {"why": "x"}}]})
    rows.append({"article_no": "no id: skipped", "neighbors": []})
    GRAPH.parent.mkdir(parents=True, exist_ok=True)
    with GRAPH.open("w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")
        f.write("\n")


def cfg_for():
    cfg = AppConfig()
    cfg.paths.law_graph_jsonl = str(GRAPH)
    return cfg


def node_out(n):
    return {"article_id": n.article_id, "graph_depth": n.graph_depth, "graph_parent": n.graph_parent,
            "relations": n.relations, "edge_conf": (n.meta or {}).get("_edge_conf"),
            "has_evidence": "_edge_evidence" in (n.meta or {})}


def mk_chunk(i, lang="en", text=None):
    return LawChunk(id=f"src.txt::{i}", law_name="Synthetic Code", article_no=f"§ {i}", article_id=str(i),
                    text=f"text of provision {i}" if text is None else text, lang=lang, source="src.txt")


class FakeStore:
    """Stands where VectorStore stands: seeded unit vectors keyed by the text."""

    def __init__(self, chunks):
        self.chunks = chunks
        rng = np.random.default_rng(99)
        self.table = {}
        for c in chunks:
            v = rng.standard_normal(DIM).astype(np.float32)
            self.table[c.text] = v / np.linalg.norm(v)
        q = rng.standard_normal(DIM).astype(np.float32)
        self.q = q / np.linalg.norm(q)

    def load(self):
        pass

    def _embed(self, texts, is_query=False):
        if isinstance(texts, str):
            return self.q.copy()
        return np.stack([self.table[t] for t in texts]).astype(np.float32)


class FakeDense:
    def __init__(self, pairs, by):
        self.pairs, self.by = pairs, by

    def search(self, query, top_k):
        return [RetrievalHit(chunk=self.by[i], score=float(s), rank=r, source="retriever", semantic_score=float(s))
                for r, (i, s) in enumerate(self.pairs[: int(top_k)], start=1)]


class FakePairs:
    def __init__(self, pairs, by):
        self.pairs, self.by = pairs, by

    def search(self, query, top_k):
        return [(self.by[i], float(s)) for i, s in self.pairs[: int(top_k)]]


class FakeReranker:
    def __init__(self, table):
        self.table = table

    def score_batch(self, query, docs):
        import re
        return [self.table[re.search(r"LawChunk\(id='([^']+)'", d).group(1)] for d in docs]

    def score(self, query, doc):
        return self.score_batch(query, [doc])[0]


def gen_hybrid_cases(chunks, store):
    """HybridRetriever.search under a routing decision, graph channel on / off."""
    rng = np.random.default_rng(31337)
    by = {c.article_id: c for c in chunks}
    pool = [c.article_id for c in chunks if c.text]
    specs = [
        ("graph_off_cut_to_seeds", dict(), 20, "GRAPH_AUGMENTED", False, (20, 20, 0)),
        ("graph_on_default", dict(), 10, "GRAPH_AUGMENTED", True, (10, 10, 10)),
        ("graph_on_enum_str", dict(graph_seed_k=4, enable_rerank=False), 10, "RoutingMode.GRAPH_AUGMENTED", True,
         (10, 10, 0)),
        ("graph_on_seed3_limit6", dict(graph_seed_k=3, graph_limit=6, min_final_score=0.0), 5, "graph_augmented", True,
         (10, 10, 0)),
        ("graph_on_but_rag_mode", dict(), 10, "RAG", True, (10, 10, 0)),
        ("graph_disabled_flag", dict(enable_graph=False), 10, "GRAPH_AUGMENTED", True, (10, 10, 0)),
    ]
    cases = []
    for name, knobs, top_k, mode, with_graph, (nd, nb, nc) in specs:
        def chan(n, lo, hi, f32):
            ids = rng.choice(pool, size=n, replace=False).tolist() if n else []
            sc = rng.uniform(lo, hi, size=n)
            sc = [float(np.float32(x)) if f32 else float(x) for x in sc]
            return sorted(zip(ids, sc), key=lambda p: -p[1])
        d, b, c = chan(nd, 0.1, 0.9, True), chan(nb, 0.0, 40.0, False), chan(nc, 5.0, 30.0, True)
        ce = {ch.id: float(x) for ch, x in zip(chunks, rng.uniform(0.0, 1.0, size=len(chunks)))}
        cfg = cfg_for()
        for k, v in knobs.items():
            object.__setattr__(cfg.retrieval, k, v)
        r = hr.HybridRetriever.__new__(hr.HybridRetriever)
        r.cfg = cfg
        r.dense, r.bm25 = FakeDense(d, by), FakePairs(b, by)
        r.colbert = FakePairs(c, by) if nc else None
        r.graph = None
        if with_graph:
            g = gr.GraphRetriever.__new__(gr.GraphRetriever)
            g.cfg, g.graph, g.store = cfg, LawGraphStore(cfg), store
            g.id2chunk = {str(x.article_id): x for x in chunks}
            r.graph = g
        fake = FakeReranker(ce)
        orig = hr.RerankerFactory.create
        hr.RerankerFactory.create = lambda self, top_k, _f=fake: _f
        try:
            res = r.search("the question", llm=None, top_k=top_k, decision=types.SimpleNamespace(mode=mode))
        finally:
            hr.RerankerFactory.create = orig
        cases.append({"name": name, "knobs": knobs, "top_k": top_k, "mode": mode, "with_graph": with_graph,
                      "dense": d, "bm25": b, "colbert": c, "ce_raw_by_id": ce,
                      "expected": [{"id": h.chunk.id, "score": float(h.score), "rank": h.rank, "source": h.source,
                                    "chunk_source": h.chunk.source, "breakdown": h.score_breakdown} for h in res]})
    return cases


def main():
    build_graph()
    out = {"walk": [], "neighbors": [], "helpers": {}, "search": []}
    cfg = cfg_for()
    gs = LawGraphStore(cfg)
    gs.load()
    out["n_nodes"] = len(gs.nodes)
    out["n_edges"] = sum(len(v) for v in gs.adj.values())

    walk_cases = [
        dict(start_ids=["0"], limit=80, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=["5", "17", "33"], limit=800, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=["5", "17", "33"], limit=7, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=["10"], limit=80, relation_max_depth={"default": 1}, rel_types=None, min_conf=0.0),
        dict(start_ids=["10"], limit=80, relation_max_depth={"default": 3, "next": 1, "prev": 1}, rel_types=None,
             min_conf=0.0),
        dict(start_ids=["22", "44"], limit=80, relation_max_depth=None, rel_types=["cite", "ref", "defined_by"],
             min_conf=0.0),
        dict(start_ids=["22", "44"], limit=80, relation_max_depth=None, rel_types=None, min_conf=0.7),
        dict(start_ids=["X1"], limit=80, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=["nope", " ", ""], limit=80, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=[" 3 ", "3", "4"], limit=0, relation_max_depth=None, rel_types=None, min_conf=0.0),
        dict(start_ids=[], limit=10, relation_max_depth=None, rel_types=None, min_conf=0.0),
    ]
    for wc in walk_cases:
        nodes = gs.walk(**wc)
        out["walk"].append({"args": wc, "nodes": [node_out(n) for n in nodes]})
    for aid, depth in [("0", 1), ("13", 2), ("59", 3), ("zz", 1), ("26", 0)]:
        out["neighbors"].append({"article_id": aid, "depth": depth,
                                 "ids": [n.article_id for n in gs.get_neighbors(aid, depth=depth)]})

    out["helpers"]["depth_decay"] = [{"depth": d, "gamma": g, "value": gr._depth_decay(d, gamma=g)}
                                     for d in (0, 1, 2, 3, 7, None) for g in (0.7, 1.0, 0.25)]
    out["helpers"]["relation_weight"] = [{"relations": r, "value": gr._relation_weight(r)}
                                         for r in ([], None, ["cite"], ["NEXT"], ["prev", "defined_by"], ["unknown"],
                                                   ["amend", "neighbor"], ["defines_term"], ["ref", "next"])]
    rng = np.random.default_rng(5)
    cs = []
    for _ in range(4):
        a, b = rng.standard_normal(DIM).astype(np.float32), rng.standard_normal(DIM).astype(np.float32)
        cs.append({"a": a.tolist(), "b": b.tolist(), "value": gr._cosine_sim(a, b)})
    cs.append({"a": [0.0] * 4, "b": [1.0, 0, 0, 0], "value": gr._cosine_sim(np.zeros(4, np.float32),
                                                                           np.array([1, 0, 0, 0], np.float32))})
    out["helpers"]["cosine_sim"] = cs

    # ---- GraphRetriever.search with the stand-in store -------------------------------------
    chunks = [mk_chunk(i, lang=("zh" if i % 9 == 0 else "en"), text=("" if i == 21 else None)) for i in range(N)
              if i not in (40, 41)]  # two graph nodes have no chunk; one chunk has empty text
    store = FakeStore(chunks)
    out["store"] = {"dim": DIM, "q": store.q.tolist(),
                    "chunks": [{"article_id": c.article_id, "lang": c.lang, "text": c.text,
                                "vec": store.table[c.text].tolist()} for c in chunks]}

    def seeds_of(ids):
        by = {c.article_id: c for c in chunks}
        return [RetrievalHit(chunk=by[i], score=1.0 - 0.01 * j, rank=j + 1) for j, i in enumerate(ids)]

    search_cases = [
        dict(seed_ids=["0"], top_k=10, lang=None, retrieval={}),
        dict(seed_ids=["5", "17", "33"], top_k=10, lang=None, retrieval={}),
        dict(seed_ids=["5", "17", "33"], top_k=50, lang="en", retrieval={}),
        dict(seed_ids=["5", "17", "33"], top_k=50, lang="zh", retrieval={}),
        dict(seed_ids=["39", "42"], top_k=5, lang=None, retrieval={"graph_limit": 12}),
        dict(seed_ids=["22", "44"], top_k=30, lang=None,
             retrieval={"graph_rel_types": ["cite", "ref", "defined_by", "next"], "graph_min_conf": 0.5,
                        "graph_depth_gamma": 1.3}),
        dict(seed_ids=["20"], top_k=30, lang=None, retrieval={"graph_walk_depths": {"default": 1}}),
        dict(seed_ids=[], top_k=10, lang=None, retrieval={}),
    ]
    for sc in search_cases:
        cfg2 = cfg_for()
        for k, v in sc["retrieval"].items():
            object.__setattr__(cfg2.retrieval, k, v)
        g = gr.GraphRetriever.__new__(gr.GraphRetriever)
        g.cfg, g.graph, g.store = cfg2, LawGraphStore(cfg2), store
        g.id2chunk = {str(c.article_id): c for c in chunks}
        hits = g.search("the question", seeds_of(sc["seed_ids"]), lang=sc["lang"], top_k=sc["top_k"])
        out["search"].append({"args": sc, "hits": [
            {"article_id": h.chunk.article_id, "score": h.score, "rank": h.rank, "source": h.source,
             "chunk_source": h.chunk.source, "score_breakdown": h.score_breakdown} for h in hits]})

    out["hybrid"] = gen_hybrid_cases(chunks, store)
    (OUT / "graph_golden.json").write_text(json.dumps(out, ensure_ascii=False, indent=1), encoding="utf-8")
    print("walk cases", len(out["walk"]), "search cases", len(out["search"]),
          "sizes", [len(w["nodes"]) for w in out["walk"]], [len(s["hits"]) for s in out["search"]])


if __name__ == "__main__":
    main()
