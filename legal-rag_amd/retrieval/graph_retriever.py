"""Graph channel: walk the law graph from the seed hits, rescore the visited articles.

Mirror of legalrag/retrieval/graph_retriever.py (GraphRetriever :53-219, helpers :19-50)
and SURVEY.md §8f-2.  The reference re-EMBEDS the text of every visited article on every
query (up to graph_limit = 800 BERT forwards, graph_retriever.py:177-179) only to take its
cosine with the query vector.  Those articles are rows of the chunk matrix that is already
resident in HBM for the dense channel (row i <-> store.chunks[i], vector_store.py:95-128),
so here the step is one `amdr_dense_score_rows` call — a row gather + dot on the device —
plus the same scalar arithmetic:

    semantic = <q, x_row> / (|q| * |x_row| + 1e-9)                 (graph_retriever.py:19-21)
    final    = semantic * (1 + depth)^-gamma * max_r w(r) * conf   (:24-46, :186-191)

Row norms are computed once per loaded index.  An article without a row in the matrix
(never the case for an index built by build_faiss_index over the same corpus) falls back
to embedding its text, as the reference does.  Results agree with the reference to fp32
rounding of the dot product (tests/test_graph.py pins walk, hydration, scores and order
against vectors produced by the reference's own code).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np

from ..schemas import LawChunk, RetrievalHit
from .graph_store import LawGraphStore
from .vector_store import VectorStore

_REL_WEIGHT = {"defined_by": 1.20, "defines_term": 1.10, "cite": 1.15, "cited": 1.15, "ref": 1.15, "amend": 1.10,
               "next": 0.95, "prev": 0.95, "neighbor": 1.00}


def _cosine_sim(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.dot(a, b) / ((np.linalg.norm(a) * np.linalg.norm(b)) + 1e-9))


def _depth_decay(depth: int, gamma: float = 0.7) -> float:
    return float(1.0 / ((1.0 + max(1, int(depth or 1))) ** gamma))


def _relation_weight(relations: List[str]) -> float:
    rels = [str(r).lower() for r in (relations or [])]
    return float(max(_REL_WEIGHT.get(r, 1.0) for r in rels)) if rels else 1.0


def _article_key(obj: Any) -> Optional[str]:
    aid = getattr(obj, "article_id", None) or getattr(obj, "id", None)
    return str(aid) if aid else None


@dataclass
class GraphRetriever:
    cfg: Any
    graph: Optional[LawGraphStore] = None
    store: Optional[VectorStore] = None
    id2chunk: Optional[Dict[str, LawChunk]] = None

    def __post_init__(self) -> None:
        if self.graph is None:
            self.graph = LawGraphStore(self.cfg)
        if self.store is None:
            self.store = VectorStore.from_config(self.cfg)
        self.store.load()
        self._bind_rows()

    def _bind_rows(self) -> None:
        """article id -> chunk and -> row of the resident matrix; the LAST chunk of an id wins,
        as in the reference's dict build (:76-80)."""
        chunks = list(getattr(self.store, "chunks", []) or [])
        self.id2chunk, self._row_of = {}, {}
        for row, c in enumerate(chunks):
            key = _article_key(c)
            if key:
                self.id2chunk[key] = c
                self._row_of[key] = row
        self._norms: Optional[np.ndarray] = None
        self._bound_index = getattr(self.store, "index", None)

    def _row_norms(self) -> Optional[np.ndarray]:
        index = getattr(self.store, "index", None)
        if index is None or not hasattr(index, "native"):
            return None
        if index is not self._bound_index:  # store reloaded (mtime guard): rows may have moved
            self._bind_rows()
        if self._norms is None:
            # a row-sharded index (vector_store.ShardedFlatIPIndex) holds rows [r0, r1) only: the norms of the other
            # rows stay -inf here and come from their own rank in _semantic's all-reduce
            r0, r1 = (index.row_offset, index.row_end) if hasattr(index, "spec") else (0, int(index.ntotal))
            norms = np.full(int(index.ntotal), -np.inf, dtype=np.float32)
            for lo in range(r0, r1, 65536):
                norms[lo:min(lo + 65536, r1)] = np.linalg.norm(index.reconstruct_n(lo, min(65536, r1 - lo)), axis=1)
            self._norms = norms
        return self._norms

    def _semantic(self, question: str, chunks: List[LawChunk], keys: List[str]) -> List[float]:
        qvec = np.asarray(self.store._embed(question), dtype=np.float32).reshape(-1)
        qn = float(np.linalg.norm(qvec))
        norms = self._row_norms()
        rows = np.array([self._row_of.get(k, -1) if norms is not None else -1 for k in keys], dtype=np.int64)
        sem = [0.0] * len(chunks)
        on_dev = np.nonzero(rows >= 0)[0]
        if on_dev.size:
            index = self.store.index
            if hasattr(index, "spec"):
                # row-sharded: every candidate row lives on exactly one rank — score the local ones, -inf for the
                # rest, ONE all-reduce(max) of (dots, norms) over the ranks; identical on every rank afterwards
                from . import sharding
                g = rows[on_dev]
                mine = (g >= index.row_offset) & (g < index.row_end)
                both = np.full((2, g.size), -np.inf, dtype=np.float32)
                if mine.any():
                    both[0, mine] = index.native.score_rows(qvec, g[mine] - index.row_offset)[0]
                    both[1, mine] = norms[g[mine]]
                both = sharding.allreduce_max_numpy(both, index._device, group=index.spec.group)
                dots, row_norms = both[0], both[1]
            else:
                dots = index.native.score_rows(qvec, rows[on_dev])[0]
                row_norms = norms[rows[on_dev]]
            for j, dot, rn in zip(on_dev, dots, row_norms):
                sem[j] = float(np.float32(dot) / np.float32(np.float32(qn * rn) + np.float32(1e-9)))
        rest = [j for j in range(len(chunks)) if rows[j] < 0]
        if rest:
            vecs = self.store._embed([chunks[j].text for j in rest])
            for j, v in zip(rest, vecs):
                sem[j] = _cosine_sim(qvec, v)
        return sem

    def search(self, question: str, seeds: List[Any], *, decision: Any = None, lang: Optional[str] = None,
               top_k: int = 10) -> List[RetrievalHit]:
        rcfg = getattr(self.cfg, "retrieval", None)
        k = max(1, int(top_k))
        depths = rcfg.graph_walk_depths if hasattr(rcfg, "graph_walk_depths") else {"default": 2}
        limit = int(getattr(rcfg, "graph_limit", k * 8) if rcfg else k * 8)
        rel_types = getattr(rcfg, "graph_rel_types", None) if rcfg else None
        min_conf = float(getattr(rcfg, "graph_min_conf", 0.0) if rcfg else 0.0)
        gamma = float(getattr(rcfg, "graph_depth_gamma", 0.7) if rcfg else 0.7)

        seed_ids = [key for key in (_article_key(getattr(h, "chunk", None)) for h in seeds or []) if key]
        if not seed_ids:
            return []
        nodes = self.graph.walk(start_ids=seed_ids, relation_max_depth=depths, limit=limit, rel_types=rel_types,
                                min_conf=min_conf)
        if not nodes:
            return []

        chunks: List[LawChunk] = []
        keys: List[str] = []
        meta: List[Dict[str, Any]] = []
        taken = set()
        for n in nodes:  # first visit of an article wins; articles without text / of another language drop out
            aid = str(getattr(n, "article_id", "") or "").strip()
            if not aid or aid in taken:
                continue
            taken.add(aid)
            c = self.id2chunk.get(aid)
            if not c or not (getattr(c, "text", "") or "").strip():
                continue
            if lang and (getattr(c, "lang", None) or "zh").strip().lower() != lang:
                continue
            cc = copy.copy(c)
            cc.source = "graph"
            chunks.append(cc)
            keys.append(aid)
            meta.append({"graph_depth": int(getattr(n, "graph_depth", 1) or 1),
                         "relations": list(getattr(n, "relations", []) or []),
                         "edge_conf": float(((getattr(n, "meta", {}) or {}).get("_edge_conf", 1.0)) or 1.0)})
        if not chunks:
            return []

        sem = self._semantic(question, chunks, keys)
        hits: List[RetrievalHit] = []
        for pos, (c, s, m) in enumerate(zip(chunks, sem, meta), start=1):
            dd, rw, conf = _depth_decay(m["graph_depth"], gamma=gamma), _relation_weight(m["relations"]), m["edge_conf"]
            final = float(s) * float(dd) * float(rw) * float(conf)
            hits.append(RetrievalHit(chunk=c, score=final, rank=pos, source="graph", score_breakdown={
                "channel": "graph", "semantic": float(s), "depth_decay": float(dd), "relation_weight": float(rw),
                "edge_conf": float(conf), "final": final, "graph_depth": m["graph_depth"], "relations": m["relations"]}))
        hits.sort(key=lambda h: float(h.score or 0.0), reverse=True)
        for r, h in enumerate(hits, start=1):
            h.rank = r
        return hits[:k]
