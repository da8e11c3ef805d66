"""Law graph: JSONL loader and the bounded breadth-first walk of the graph channel.

Host-side mirror of legalrag/retrieval/graph_store.py (LawGraphStore: load :21-87,
walk :89-169, get_neighbors :171-196, get_node :198).  The walk is pointer chasing over
a few hundred nodes per query — host work by nature; what the reference spends its time
on in this channel, re-embedding every visited article, is replaced by a gather + dot on
the resident chunk matrix in graph_retriever.py.

Behaviour kept, including the parts that look accidental:
  * a node line needs `article_id` or `id`; neighbours may be bare strings (relation
    "neighbor", conf 1.0) or dicts keyed `article_id` / `id`; `conf` 0 / missing -> 1.0;
  * walk() marks a neighbour visited BEFORE checking that it exists as a node, visits
    neighbours in file order, stops at `limit` results, and limits depth PER RELATION of the
    edge that reached the current node (`relation_max_depth[rel]`, else "default", else 2);
  * returned nodes are shallow copies carrying graph_depth / graph_parent / relations=[rel];
    `_edge_conf` and `_edge_evidence` are attached to meta ONLY when the edge has evidence —
    so an edge's confidence reaches the score only for edges with evidence.
"""
from __future__ import annotations

import copy
import json
import logging
from collections import defaultdict, deque
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

from ..schemas import LawNode, Neighbor

logger = logging.getLogger(__name__)
Edge = Tuple[str, str, float, Optional[Dict[str, Any]]]


def _clean(x: Any) -> str:
    return str(x or "").strip()


class LawGraphStore:
    def __init__(self, cfg):
        self.cfg = cfg
        self.graph_path = Path(getattr(cfg.paths, "law_graph_jsonl"))
        self.nodes: Dict[str, LawNode] = {}
        self.adj: Dict[str, List[Edge]] = defaultdict(list)
        self._loaded = False

    # ------------------------------------------------------------------ load
    @staticmethod
    def _parse_neighbors(raw) -> List[Neighbor]:
        out: List[Neighbor] = []
        for nb in raw or []:
            if isinstance(nb, str):
                out.append(Neighbor(article_id=str(nb), relation="neighbor", conf=1.0))
            elif isinstance(nb, dict):
                dst = _clean(nb.get("article_id") or nb.get("id"))
                if dst:
                    out.append(Neighbor(article_id=dst, relation=str(nb.get("relation") or "neighbor"),
                                        conf=float(nb.get("conf", 1.0) or 1.0), evidence=nb.get("evidence")))
        return out

    def load(self) -> None:
        if self._loaded:
            return
        if not self.graph_path.exists():
            raise FileNotFoundError(f"Graph JSONL not found: {self.graph_path}")
        nodes: Dict[str, LawNode] = {}
        with self.graph_path.open("r", encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                obj = json.loads(line)
                aid = _clean(obj.get("article_id") or obj.get("id"))
                if not aid:
                    continue
                nodes[aid] = LawNode(article_id=aid, article_no=str(obj.get("article_no") or ""),
                                     law_name=obj.get("law_name"), title=obj.get("title"), chapter=obj.get("chapter"),
                                     section=obj.get("section"), neighbors=self._parse_neighbors(obj.get("neighbors")),
                                     meta=obj.get("meta") or {})
        adj: Dict[str, List[Edge]] = defaultdict(list)
        n_edges = 0
        for src, node in nodes.items():
            for e in node.neighbors:
                adj[src].append((e.article_id, e.relation, float(e.conf or 1.0), e.evidence))
                n_edges += 1
        self.nodes, self.adj, self._loaded = nodes, adj, True
        logger.info("[Graph] Loaded %d law nodes, %d edges", len(nodes), n_edges)

    # ------------------------------------------------------------------ walk
    def walk(self, start_ids: List[str], limit: int = 80, relation_max_depth: Optional[Dict[str, int]] = None,
             rel_types: Optional[List[str]] = None, min_conf: float = 0.0) -> List[LawNode]:
        self.load()
        start = [_clean(x) for x in (start_ids or []) if _clean(x)]
        if not start:
            return []
        rcfg = self.cfg.retrieval
        if relation_max_depth is None:
            relation_max_depth = getattr(rcfg, "graph_walk_depths", None) or {"default": 2}
        if rel_types is None:
            rel_types = getattr(rcfg, "graph_rel_types", None) if rcfg else None
        default_depth = relation_max_depth.get("default", 2)
        limit = max(1, int(limit))
        allowed = {str(r) for r in rel_types} if rel_types else None
        min_conf = float(min_conf or 0.0)

        seen = set(start)
        queue: deque = deque((sid, 0, None) for sid in start)  # (node, hops from a seed, relation that led here)
        found: List[LawNode] = []
        while queue and len(found) < limit:
            cur, dist, via = queue.popleft()
            if dist >= (relation_max_depth.get(via, default_depth) if via else default_depth):
                continue
            for dst, rel, conf, evidence in self.adj.get(cur, []):
                if (min_conf > 0 and conf < min_conf) or (allowed is not None and rel not in allowed) or dst in seen:
                    continue
                seen.add(dst)
                stored = self.nodes.get(dst)
                if not stored:
                    continue
                node = copy.copy(stored)
                node.graph_depth, node.graph_parent, node.relations = dist + 1, cur, [rel]
                if evidence:
                    node.meta = dict(node.meta or {})
                    node.meta["_edge_evidence"] = evidence
                    node.meta["_edge_conf"] = conf
                found.append(node)
                if len(found) >= limit:
                    break
                queue.append((dst, dist + 1, rel))
        return found

    def get_neighbors(self, article_id: str, depth: int = 1) -> List[LawNode]:
        self.load()
        root = str(article_id).strip()
        if root not in self.nodes:
            return []
        seen, frontier, out = {root}, [root], []
        for _ in range(max(1, int(depth))):
            nxt: List[str] = []
            for aid in frontier:
                for dst, _rel, _conf, _ev in self.adj.get(aid, []):
                    if dst in seen:
                        continue
                    seen.add(dst)
                    node = self.nodes.get(dst)
                    if node:
                        out.append(node)
                        nxt.append(dst)
            frontier = nxt
        return out

    def get_node(self, article_id: str) -> Optional[LawNode]:
        return self.nodes.get(article_id)
