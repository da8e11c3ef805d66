"""Host-side pieces of the product (no GPU): BM25 statistics, config layout,
rerank utilities, dedup, evaluation metrics — against the oracle / goldens."""
import copy
import math

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from legal_rag_amd import evaluation
from legal_rag_amd.bm25_model import BM25Okapi
from legal_rag_amd.config import AppConfig
from legal_rag_amd.retrieval import rerankers
from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
from legal_rag_amd.schemas import LawChunk, RetrievalHit
from legal_rag_amd.text import tokenize_en
from oracle import bm25 as OB

UTIL = load_golden("util_golden.json")


@pytest.fixture(scope="module")
def ucc():
    return load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")


def test_corpus_statistics_regression(ucc):
    """SURVEY.md §8c(5) / BASELINE.md §2: 592 parsed, 591 unique (dup id ucc_4A.txt::4A-102)."""
    raw = [l for l in (GOLDEN / "corpus" / "law_en.jsonl").read_text(encoding="utf-8").splitlines() if l.strip()]
    assert len(raw) == 592 and len(ucc) == 591
    import json
    ids = [json.loads(l)["id"] for l in raw]
    assert sorted({i for i in ids if ids.count(i) > 1}) == ["ucc_4A.txt::4A-102"]
    toks_all = [tokenize_en(json.loads(l)["text"]) for l in raw]
    assert sum(len(t) for t in toks_all) == 148124
    bm = BM25Okapi(toks_all)
    assert len(bm.idf) == 3926 and sum(len(d) for d in bm.doc_freqs) == 54010 and max(bm.doc_len) == 4550
    zh = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_zh.jsonl")
    assert len(zh) == 1260


def test_product_bm25_statistics_equal_oracle(ucc):
    toks = [tokenize_en(c.text) for c in ucc]
    a, b = BM25Okapi(toks), OB.BM25Okapi(toks)
    assert a.avgdl == b.avgdl and a.average_idf == b.average_idf and a.corpus_size == b.corpus_size
    assert list(a.idf.items()) == list(b.idf.items())  # same order, bit-equal values
    ca, cb = a.to_csr(), OB.to_csr(b)
    for x, key in zip(ca, ("term_ptr", "post_doc", "post_tf", "idf", "doc_len")):
        assert np.array_equal(x, cb[key]), key
    assert a.term_ids(["the", "Buyer", "zzz"]) == [b_ for b_ in (cb["vocab"]["the"], -1, -1)]


def test_negative_idf_floor():
    docs = [["a", "b"], ["a"], ["a", "c"], ["a"]]
    bm = BM25Okapi(docs)
    raw = math.log(4 - 4 + 0.5) - math.log(4 + 0.5)
    assert raw < 0 and bm.idf["a"] == 0.25 * bm.average_idf
    assert bm.idf == OB.BM25Okapi(docs).idf


def test_config_layout_and_registry(tmp_path):
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    r = cfg.retrieval
    root = tmp_path / "index" / "en"
    assert r.faiss_index_file == str(root / "faiss" / "faiss.index")
    assert r.faiss_meta_file == str(root / "faiss" / "faiss_meta.jsonl")
    assert r.bm25_index_file == str(root / "bm25.pkl")
    assert r.colbert_meta_file == str(root / "colbert" / "colbert_meta.jsonl")
    assert r.embedding_model == r.embedding_model_en and r.colbert_index_name == "law_en"
    assert not root.exists()  # constructing a config creates nothing
    (root / "versions" / "v2").mkdir(parents=True)
    (root / "ACTIVE").write_text("v2")
    assert cfg.with_lang("en").retrieval.bm25_index_file == str(root / "versions" / "v2" / "bm25.pkl")
    (root / "ACTIVE").write_text("missing")
    assert cfg.with_lang("en").retrieval.bm25_index_file == str(root / "bm25.pkl")
    zh = cfg.with_lang("zh")
    assert zh.retrieval.embedding_model == zh.retrieval.embedding_model_zh and "index/zh" in zh.retrieval.bm25_index_file
    # reference defaults of the knobs the path reads (legalrag/config.py:92-129)
    assert (r.top_k, r.dense_weight, r.bm25_weight, r.colbert_weight, r.min_final_score) == (10, 0.6, 0.4, 0.35, 0.2)
    assert (r.rerank_top_n, r.rrf_alpha, r.rerank_beta, r.rrf_k, r.fusion_method) == (30, 0.5, 0.35, 60, "rrf_norm_blend")


def test_rerank_utils_match_reference_vectors():
    for row in UTIL["minmax"]:
        assert rerankers.minmax_normalize(row["in"]) == row["rerank_minmax"]
    rn = UTIL["rerank_norm"]
    assert [rerankers.sigmoid(x) for x in rn["x"]] == rn["sigmoid"]
    assert rerankers.sigmoid_calibrate(rn["x"], 0.25) == rn["calibrate_t0p25"]
    assert rerankers.sigmoid_calibrate(rn["x"], 0.0) == rn["calibrate_t0"]

    class Fake:
        def __init__(self, t):
            self.t = t

        def score_batch(self, q, docs):
            return [self.t[d] for d in docs]
    for case in UTIL["rerank_candidates"]:
        cands = [{"text": f"doc {i}", "id": i} for i in range(9)]
        res = rerankers.rerank_candidates("q", cands, Fake(case["raw_by_text"]), top_n=case["top_n"],
                                          normalize=case["normalize"], sigmoid_temperature=0.5, include_debug=True)
        got = [{"id": c["id"], "raw": r.raw_score, "norm": r.norm_score, "meta": r.meta} for c, r in res]
        assert got == case["expected"]


def test_rerank_doc_text_quirk():
    """Hits are neither str nor dict -> the cross-encoder sees str(hit) (rerankers.py:78-86)."""
    c = LawChunk(id="a::1", law_name="L", article_no="§ 1", article_id="1", text="T", lang="en")
    h = RetrievalHit(chunk=c, score=0.5, rank=1)
    s = rerankers._to_doc_text(h)
    assert s.startswith("chunk=LawChunk(id='a::1', law_name='L', chapter=None, section=None, article_no='§ 1', "
                        "article_id='1', text='T', lang='en', source=None, start_char=None, end_char=None) score=0.5 "
                        "rank=1 source='retriever'")
    assert rerankers._to_doc_text({"content": "x"}) == "x" and rerankers._to_doc_text("y") == "y"


def test_dedup_keep_best_matches_reference_vectors():
    from legal_rag_amd.retrieval.hybrid_retriever import _dedup_keep_best
    for row in UTIL["dedup"]:
        hits = []
        for h in copy.deepcopy(row["in"]):
            i = h["id"].split("::")[1]
            c = LawChunk(id=h["id"], law_name="Synthetic Code", article_no=f"§ {i}", article_id=i, text="t", lang="en")
            hits.append(RetrievalHit(chunk=c, score=h["score"], rank=h["rank"], score_breakdown=h["breakdown"]))
        got = _dedup_keep_best(hits)
        exp = row["expected"]
        assert [g.chunk.id for g in got] == [e["id"] for e in exp]
        for g, e in zip(got, exp):
            assert g.score == e["score"] and g.rank == e["rank"]
            gb, eb = g.score_breakdown or {}, e["breakdown"] or {}
            assert gb.get("channel_contrib") == eb.get("channel_contrib")
            assert sorted(gb.get("channel", [])) == sorted(eb.get("channel", []))


def test_metrics():
    pred = ["a", "b", "c", "d"]
    assert evaluation.recall_at_k(pred, {"c"}, 3) == 1.0 and evaluation.recall_at_k(pred, {"c"}, 2) == 0.0
    assert evaluation.mrr_at_k(pred, {"b"}, 10) == 0.5 and evaluation.hit_at_k(pred, {"z"}, 10) == 0.0
    assert abs(evaluation.ndcg_at_k(pred, {"b"}, 10) - 1 / math.log2(3)) < 1e-15
    assert evaluation.recall_at_k(pred, set(), 5) == 0.0


def test_synthetic_queries_are_deterministic(ucc):
    a = evaluation.synthetic_queries(ucc, seed=0)
    assert a == evaluation.synthetic_queries(ucc, seed=0) and len(a) == 1168
    assert a[0] == ("Short Titles", "1-101", "title")


def test_colbert_from_config_does_not_deadlock(tmp_path, monkeypatch):
    """from_config() builds the instance under the registry lock and the instance fills the
    searcher cache under the same lock: it must be re-entrant (regression: this hung a GPU run)."""
    import threading

    from legal_rag_amd import _native
    from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index
    from legal_rag_amd.retrieval.colbert_retriever import ColBERTRetriever

    class FakeMaxSim:
        def __init__(self, D, doc_ptr, device=0):
            self.n_docs = len(doc_ptr) - 1

    monkeypatch.setattr(_native, "MaxSimIndex", FakeMaxSim)
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    cfg.retrieval.encoder_backend = "hashing"
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")[:5]
    out = build_colbert_index(cfg, chunks)
    assert (out / "amdr_tokens.npz").exists()
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("r", ColBERTRetriever.from_config(cfg)), daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "ColBERTRetriever.from_config deadlocked"
    r = box["r"]
    assert r.enabled and r._searcher.n_docs == 5 and ColBERTRetriever.from_config(cfg) is r
    assert r.search("   ", 3) == []


def test_llm_rerankers_and_factory_choice():
    import asyncio

    class FakeLLM:
        def __init__(self):
            self.calls = 0

        def chat(self, messages, tag=None):
            self.calls += 1
            doc = messages[1]["content"]
            if "alpha" in doc:
                return '{"score": 0.9, "reason": "direct"}'
            if "beta" in doc:
                return "I would say 0.25 maybe"
            if "gamma" in doc:
                return '{"score": 7}'
            return "no idea"

    llm = FakeLLM()
    r = rerankers.LLMReranker(llm=llm)
    assert r.score_batch("q", ["alpha text", "beta text", "gamma", "delta"]) == [0.9, 0.25, 1.0, 0.0]
    assert "Candidate provision:" in rerankers.build_llm_rerank_prompt("q", "d")
    c = rerankers.CachedLLMReranker(llm=llm)
    n0 = llm.calls
    assert c.score("q", "alpha") == 0.9 and c.score("q", "alpha") == 0.9 and llm.calls == n0 + 1
    f = rerankers.RerankerFactory(llm=llm, cross_model="missing-model", llm_threshold=30)
    assert isinstance(f.create(10), rerankers.CachedLLMReranker)      # llm given and top_k <= threshold
    with pytest.raises(RuntimeError):
        f.create(31)                                                   # falls to the cross-encoder: not local -> loud
    a = rerankers.AsyncCachedLLMReranker(llm=llm, max_concurrency=2)
    got = asyncio.run(a.score_batch("q", ["alpha", "beta", "alpha"]))
    assert got == [0.9, 0.25, 0.9]
    long_q = "x" * 1000
    assert len(r._call_llm(long_q, "alpha")) > 0


# ---- /retrieve service contract (retrieval_api.py:51-77), vectors from the reference's own route ----
def test_retrieve_response_matches_reference_route():
    from types import SimpleNamespace
    from legal_rag_amd import service
    from legal_rag_amd.schemas import LawChunk, RetrievalHit
    cases = load_golden("service_golden.json")["cases"]
    assert len(cases) >= 10

    def mk_hit(i, score, rank):
        c = LawChunk(id=f"src.txt::{i}", law_name="Synthetic Code", article_no=f"§ {i}", article_id=str(i),
                     text=f"text of provision {i}", lang="en", source="src.txt")
        return RetrievalHit(chunk=c, score=score, rank=rank, source="retriever",
                            score_breakdown={"channel": ["dense"], "dense_raw": score})

    class Retriever:
        def __init__(self):
            self.calls = []

        def search(self, question, top_k=10, decision=None):
            self.calls.append({"question": question, "top_k": top_k, "mode": decision.mode})
            return [mk_hit(i, 1.0 - 0.01 * i, i + 1) for i in range(min(top_k, 4))]

    for case in cases:
        if case.get("not_ready"):
            with pytest.raises(service.ServiceError) as e:
                service.retrieve_response(case["body"], None, None, None)
            assert (e.value.status_code, e.value.detail) == (503, case["http_error"]["detail"])
            continue
        cfg = SimpleNamespace(retrieval=SimpleNamespace(top_k=case["cfg_top_k"]))
        dec = SimpleNamespace(top_k_factor=case["top_k_factor"], mode=case["mode"])
        dec.model_dump = lambda d=dec: case["response"]["decision"] if "response" in case else {}
        router = SimpleNamespace(route=lambda q, d=dec: d)
        r = Retriever()
        if "http_error" in case:
            with pytest.raises(service.ServiceError) as e:
                service.retrieve_response(case["body"], r, router, cfg)
            assert e.value.status_code == case["http_error"]["status_code"]
            assert e.value.detail == case["http_error"]["detail"]
            assert r.calls == []
            continue
        got = service.retrieve_response(case["body"], r, router, cfg)
        assert got == case["response"], case["body"]
        assert r.calls == case["search_calls"]
    # the same clamp as RagPipeline.retrieve (rag_pipeline.py:249-251)
    assert [service.effective_top_k(k, f) for k, f in ((10, 1.0), (1, 1.0), (25, 2.0), (5, 0.5), (9, 1.3))] == \
        [10, 3, 30, 3, 11]
    route = service.make_route(Retriever(), SimpleNamespace(route=lambda q: SimpleNamespace(top_k_factor=1.0)),
                               SimpleNamespace(retrieval=SimpleNamespace(top_k=10)), http_exception=None)
    with pytest.raises(service.ServiceError):
        route({})


def test_hit_text_equals_pydantic_str():
    """HybridRetriever._hit_text(h) == str(h) — the string the reference's rerank stage hands to the cross-encoder — for
    hits of every shape the product builds (validated and model_construct-ed, with / without breakdowns, graph hits,
    non-ASCII text, quotes and backslashes in the text)."""
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    from legal_rag_amd.schemas import LawChunk, RetrievalHit
    chunks = [LawChunk(id=f"f::{i}", law_name="UCC", article_no=f"2-{i}", article_id=f"2-{i}",
                       text=t, lang=lang, source=src, start_char=i, end_char=i + 9, chapter=ch)
              for i, (t, lang, src, ch) in enumerate([
                  ("A merchant's \"warranty\" \\ of fitness.\n(a) line", "en", "ucc_2.txt", None),
                  ("认购书或者订购书等是否属于预约合同？", "zh", None, "第二章"),
                  ("", None, None, None)])]
    sb = {"fusion_method": "rrf_norm_blend", "rrf_k": 60, "alpha": 0.5, "channel_weights": {"dense": 0.6, "bm25": 0.4},
          "channel": ["dense", "bm25"], "channel_contrib": {"dense": 0.3, "bm25": 0.1, "colbert": 0.0}, "rrf_norm": 1.0,
          "weighted_sum": 0.4, "dense_norm": 0.5, "bm25_norm": 0.25, "colbert_norm": 0.0, "zh_exact": False}
    hits = [RetrievalHit(chunk=chunks[0], score=0.8, rank=1, score_breakdown=sb),
            RetrievalHit.model_construct(chunk=chunks[1], score=1e-17, rank=2, source="retriever", score_breakdown=dict(sb)),
            RetrievalHit(chunk=chunks[2], score=float("inf"), rank=None),
            RetrievalHit(chunk=chunks[0], score=0.25, rank=4, source="graph", graph_depth=2, relations=["cites", "x"],
                         seed_article_id="2-1", semantic_score=0.33, score_breakdown={"graph": {"decay": 0.5}}),
            RetrievalHit(chunk=chunks[1], score=-0.0, rank=5, source="rerank",
                         score_breakdown={"rerank_raw": 0.1, "rerank_norm": 1.0, "rerank_beta": 0.35})]
    for h in hits * 2:  # second pass: the cached chunk reprs
        assert HybridRetriever._hit_text(h) == str(h)
    assert HybridRetriever._hit_text("plain") == "plain" and HybridRetriever._hit_text({"text": "t"}) == "t"
