"""Graph channel: LawGraphStore.walk / GraphRetriever.search against
vectors produced by the REFERENCE'S OWN CODE (tests/golden/gen_graph_golden.py ran
legalrag/retrieval/graph_store.py + graph_retriever.py on the synthetic graph fixture)."""
import json
import types

import numpy as np
import pytest

from conftest import GOLDEN

G = json.loads((GOLDEN / "graph_golden.json").read_text(encoding="utf-8"))
GRAPH_FILE = GOLDEN / "graph" / "law_graph_fixture.jsonl"


def make_cfg(**retrieval):
    from legal_rag_amd.config import AppConfig
    cfg = AppConfig()
    cfg.paths.law_graph_jsonl = str(GRAPH_FILE)
    for k, v in retrieval.items():
        setattr(cfg.retrieval, k, v)
    return cfg


def node_view(n):
    return {"article_id": n.article_id, "graph_depth": n.graph_depth, "graph_parent": n.graph_parent,
            "relations": n.relations, "edge_conf": (n.meta or {}).get("_edge_conf"),
            "has_evidence": "_edge_evidence" in (n.meta or {})}


def test_graph_load_counts():
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    gs = LawGraphStore(make_cfg())
    gs.load()
    assert len(gs.nodes) == G["n_nodes"]
    assert sum(len(v) for v in gs.adj.values()) == G["n_edges"]
    assert gs.get_node("X1") is not None and gs.get_node("nope") is None


def test_graph_missing_file_raises(tmp_path):
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    cfg = make_cfg()
    cfg.paths.law_graph_jsonl = str(tmp_path / "absent.jsonl")
    with pytest.raises(FileNotFoundError):
        LawGraphStore(cfg).load()


@pytest.mark.parametrize("case", G["walk"], ids=[f"walk{i}" for i in range(len(G["walk"]))])
def test_walk_matches_reference(case):
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    gs = LawGraphStore(make_cfg())
    got = [node_view(n) for n in gs.walk(**case["args"])]
    assert got == case["nodes"]


def test_walk_does_not_touch_stored_nodes():
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    gs = LawGraphStore(make_cfg())
    gs.walk(["5", "17"], limit=100)
    assert all(n.graph_depth is None and n.relations is None and "_edge_conf" not in n.meta for n in gs.nodes.values())


@pytest.mark.parametrize("case", G["neighbors"], ids=[c["article_id"] + "_" + str(c["depth"]) for c in G["neighbors"]])
def test_get_neighbors_matches_reference(case):
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    gs = LawGraphStore(make_cfg())
    assert [n.article_id for n in gs.get_neighbors(case["article_id"], depth=case["depth"])] == case["ids"]


def test_helpers_match_reference():
    from legal_rag_amd.retrieval import graph_retriever as gr
    for c in G["helpers"]["depth_decay"]:
        assert gr._depth_decay(c["depth"], gamma=c["gamma"]) == c["value"]
    for c in G["helpers"]["relation_weight"]:
        assert gr._relation_weight(c["relations"]) == c["value"]
    for c in G["helpers"]["cosine_sim"]:
        got = gr._cosine_sim(np.array(c["a"], np.float32), np.array(c["b"], np.float32))
        assert abs(got - c["value"]) <= 1e-6


# ----------------------------------------------------------------------------- retriever
def store_chunks():
    from legal_rag_amd.schemas import LawChunk
    out = []
    for c in G["store"]["chunks"]:
        out.append(LawChunk(id=f"src.txt::{c['article_id']}", law_name="Synthetic Code", article_no=f"§ {c['article_id']}",
                            article_id=c["article_id"], text=c["text"], lang=c["lang"], source="src.txt"))
    return out


class HostStore:
    """VectorStore stand-in without a device index: GraphRetriever must then embed the texts."""

    def __init__(self):
        self.chunks = store_chunks()
        self.table = {c["text"]: np.array(c["vec"], np.float32) for c in G["store"]["chunks"]}
        self.q = np.array(G["store"]["q"], np.float32)
        self.embedded = 0

    def load(self):
        pass

    def _embed(self, texts, is_query=False):
        if isinstance(texts, str):
            return self.q.copy()
        self.embedded += len(texts)
        return np.stack([self.table[t] for t in texts]).astype(np.float32)


def seeds_of(store, ids):
    from legal_rag_amd.schemas import RetrievalHit
    by = {c.article_id: c for c in store.chunks}
    return [RetrievalHit(chunk=by[i], score=1.0 - 0.01 * j, rank=j + 1) for j, i in enumerate(ids)]


def check_search(store, tol):
    from legal_rag_amd.retrieval.graph_retriever import GraphRetriever
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    for case in G["search"]:
        a = case["args"]
        cfg = make_cfg(**a["retrieval"])
        g = GraphRetriever(cfg, graph=LawGraphStore(cfg), store=store)
        hits = g.search("the question", seeds_of(store, a["seed_ids"]), lang=a["lang"], top_k=a["top_k"])
        exp = case["hits"]
        assert [h.chunk.article_id for h in hits] == [e["article_id"] for e in exp], a
        for h, e in zip(hits, exp):
            assert h.rank == e["rank"] and h.source == e["source"] == "graph" and h.chunk.source == e["chunk_source"]
            assert abs(h.score - e["score"]) <= tol
            sb, eb = h.score_breakdown, e["score_breakdown"]
            assert set(sb) == set(eb)
            for key in ("channel", "graph_depth", "relations", "depth_decay", "relation_weight", "edge_conf"):
                assert sb[key] == eb[key], key
            assert abs(sb["semantic"] - eb["semantic"]) <= tol and abs(sb["final"] - eb["final"]) <= tol
        # the stored chunks are not relabelled
        assert all(c.source == "src.txt" for c in store.chunks)


def test_graph_retriever_matches_reference_on_host_store():
    store = HostStore()
    check_search(store, 1e-6)
    assert store.embedded > 0  # no device index: texts were embedded, as the reference does


def test_hybrid_search_graph_relabels_and_swallows():
    from legal_rag_amd.retrieval.graph_retriever import GraphRetriever
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    store = HostStore()
    cfg = make_cfg()
    hr = HybridRetriever.__new__(HybridRetriever)
    hr.cfg = cfg
    hr.graph = GraphRetriever(cfg, graph=LawGraphStore(cfg), store=store)
    hits = hr.search_graph("q", 5, seeds=seeds_of(store, ["5", "17", "33"]))
    assert len(hits) == 5 and [h.rank for h in hits] == [1, 2, 3, 4, 5]
    assert all(h.source == "retriever" and h.score_breakdown["channel"] == ["graph"] for h in hits)
    assert hits == sorted(hits, key=lambda h: -h.score)
    hr.graph = types.SimpleNamespace(search=lambda *a, **k: 1 / 0)
    assert hr.search_graph("q", 5, seeds=[]) == []
    hr.graph = None
    assert hr.search_graph("q", 5, seeds=[]) == []


@pytest.mark.gpu
def test_graph_retriever_rescoring_on_device():
    """Same vectors, but the store holds the chunk matrix on the GPU: the semantic term comes
    from amdr_dense_score_rows (row gather + dot) and nothing is re-embedded."""
    from legal_rag_amd.retrieval.vector_store import FlatIPIndex

    class DeviceStore(HostStore):
        def __init__(self):
            super().__init__()
            X = np.stack([self.table[c.text] for c in self.chunks]).astype(np.float32)
            self.index = FlatIPIndex(X, device=0)

    store = DeviceStore()
    check_search(store, 2e-6)
    assert store.embedded == 0
