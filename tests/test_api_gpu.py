"""The reference-shaped Python API on the GPU: builders -> artifacts ->
HybridRetriever.search / _fuse / search_batch, against the oracle and against
vectors produced by the reference's own HybridRetriever code."""
import copy
import logging

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from helpers import assert_hits_equal_mod_ties

pytestmark = pytest.mark.gpu

FUSE = load_golden("fusion_golden.json")["cases"]
SEARCH = load_golden("search_golden.json")["cases"]


def dump(h):
    return {"id": h.chunk.id, "score": float(h.score), "rank": h.rank, "source": h.source,
            "breakdown": h.score_breakdown}


def mk_chunk(i):
    from legal_rag_amd.schemas import LawChunk
    return LawChunk(id=f"src.txt::{i}", law_name="Synthetic Code", article_no=f"§ {i}", article_id=str(i),
                    text=f"text of provision {i}", lang="en", source="src.txt")


def bare_retriever(**knobs):
    """HybridRetriever without indexes (channels injected), like the golden generator."""
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg = AppConfig()
    for k, v in knobs.items():
        setattr(cfg.retrieval, k, v)
    r = HybridRetriever.__new__(HybridRetriever)
    r.cfg = cfg
    r.dense = r.bm25 = r.colbert = r.graph = None
    return r


def hits_from(pairs, chunks, ch):
    from legal_rag_amd.schemas import RetrievalHit
    return [RetrievalHit(chunk=chunks[i], score=float(s), rank=r, source="retriever",
                         score_breakdown={"channel": [ch], f"{ch}_raw": float(s)})
            for r, (i, s) in enumerate(pairs, start=1)]


@pytest.mark.parametrize("case", FUSE, ids=[c["name"] for c in FUSE])
def test_product_fuse_matches_reference_vectors(case):
    chunks = {i: mk_chunk(i) for i in range(400)}
    r = bare_retriever(**case["knobs"])
    out = r._fuse(dense_hits=hits_from(case["dense"], chunks, "dense"), bm25_hits=hits_from(case["bm25"], chunks, "bm25"),
                  colbert_hits=hits_from(case["colbert"], chunks, "colbert"))
    assert_hits_equal_mod_ties([dump(h) for h in out], case["expected"])


class FakeDense:
    def __init__(self, pairs, chunks):
        self.pairs, self.chunks = pairs, chunks

    def search(self, query, top_k):
        from legal_rag_amd.schemas import RetrievalHit
        return [RetrievalHit(chunk=self.chunks[i], score=float(s), rank=r, source="retriever", semantic_score=float(s))
                for r, (i, s) in enumerate(self.pairs[: int(top_k)], start=1)]


class FakePairs:
    def __init__(self, pairs, chunks):
        self.pairs, self.chunks = pairs, chunks

    def search(self, query, top_k):
        return [(self.chunks[i], float(s)) for i, s in self.pairs[: int(top_k)]]


class FakeReranker:
    def __init__(self, table):
        self.table, self.seen = table, []

    def score_batch(self, query, docs):
        import re
        self.seen.extend(docs)
        return [self.table[re.search(r"LawChunk\(id='([^']+)'", d).group(1)] for d in docs]


@pytest.mark.parametrize("case", SEARCH, ids=[c["name"] for c in SEARCH])
def test_product_search_matches_reference_vectors(case, monkeypatch):
    """Whole HybridRetriever.search() (channels faked exactly as in the generator):
    same hits, scores, ranks, sources, breakdowns — and the SAME strings handed to
    the cross-encoder (the str(hit) quirk)."""
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    chunks = {i: mk_chunk(i) for i in range(300)}
    r = bare_retriever(**case["knobs"])
    r.dense = FakeDense(case["dense"], chunks)
    r.bm25 = FakePairs(case["bm25"], chunks)
    r.colbert = FakePairs(case["colbert"], chunks) if case["colbert"] else None
    fake = FakeReranker(case["ce_raw_by_id"])
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: fake)
    out = r.search("synthetic question", llm=None, top_k=case["top_k"], decision=None)
    assert_hits_equal_mod_ties([dump(h) for h in out], case["expected"])
    if fake.seen != case["ce_docs_seen"]:
        # exactly tied fused scores may swap places (set-order in the reference): the same
        # strings up to the rank number inside them, in the same order up to those swaps
        import re

        def norm(xs):
            return sorted(re.sub(r" rank=\d+ ", " rank=? ", s) for s in xs)
        assert norm(fake.seen) == norm(case["ce_docs_seen"])
        scores_g = [re.search(r" score=(\S+) ", s).group(1) for s in fake.seen]
        scores_e = [re.search(r" score=(\S+) ", s).group(1) for s in case["ce_docs_seen"]]
        assert scores_g == scores_e


GRAPH = load_golden("graph_golden.json")


@pytest.mark.parametrize("case", GRAPH["hybrid"], ids=[c["name"] for c in GRAPH["hybrid"]])
def test_product_search_with_routing_decision_matches_reference(case, monkeypatch):
    """HybridRetriever.search under a routing decision (graph channel on / off / not selected):
    seeds cut, graph hits appended, rerank, dedup — against the reference's own search()."""
    import types
    from test_graph import GRAPH_FILE, HostStore
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    from legal_rag_amd.retrieval.graph_retriever import GraphRetriever
    from legal_rag_amd.retrieval.graph_store import LawGraphStore
    store = HostStore()
    by = {c.article_id: c for c in store.chunks}
    r = bare_retriever(**case["knobs"])
    r.cfg.paths.law_graph_jsonl = str(GRAPH_FILE)
    r.dense = FakeDense(case["dense"], by)
    r.bm25 = FakePairs(case["bm25"], by)
    r.colbert = FakePairs(case["colbert"], by) if case["colbert"] else None
    if case["with_graph"]:
        r.graph = GraphRetriever(r.cfg, graph=LawGraphStore(r.cfg), store=store)
    fake = FakeReranker(case["ce_raw_by_id"])
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: fake)
    out = r.search("the question", llm=None, top_k=case["top_k"], decision=types.SimpleNamespace(mode=case["mode"]))
    assert_hits_equal_mod_ties([dump(h) for h in out], case["expected"], float_tol=1e-6)
    assert [h.chunk.source for h in out] == [e["chunk_source"] for e in case["expected"]]


# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ucc_index(tmp_path_factory):
    """Build the UCC-en indexes with the product builders (stand-in encoders)."""
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index
    from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    data = tmp_path_factory.mktemp("data")
    cfg = AppConfig.for_data_dir(str(data), "en")
    cfg.retrieval.encoder_backend = "hashing"
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")[:200]
    build_faiss_index(cfg, chunks)
    build_bm25_index(cfg, chunks)
    build_colbert_index(cfg, chunks)
    return cfg, chunks


def oracle_channels(cfg, chunks, question, k):
    from legal_rag_amd import encoders, text
    from oracle import bm25 as OB
    from oracle import dense as OD
    from oracle import maxsim as OM
    emb = encoders.HashingEmbedder(768)
    X = emb.encode([c.text for c in chunks])
    q = emb.encode_queries([question])
    ds, di = OD.flatip_topk(X, q, k)
    ob = OB.BM25Okapi([OB.tokenize_en(c.text) for c in chunks])
    b = OB.search(ob, text.jieba_cut(question), k)
    te = encoders.HashingTokenEmbedder()
    mats = [te.encode_doc(c.text.strip()) for c in chunks]
    ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])])
    cs, ci = OM.maxsim_topk(te.encode_query(question.strip())[None], np.concatenate(mats), ptr, k)
    return ([(chunks[i].id, float(s)) for s, i in zip(ds[0], di[0]) if i >= 0],
            [(chunks[i].id, s) for i, s in b],
            [(chunks[i].id, float(s)) for s, i in zip(cs[0], ci[0]) if i >= 0])


QUESTIONS = ["what warranty does a merchant give that goods are merchantable",
             "Short Titles", "statute of frauds signed writing sale of goods price of $500",
             "What is § 2-314?", "risk of loss passes to the buyer on tender of delivery"]


@pytest.mark.parametrize("question", QUESTIONS)
def test_end_to_end_search_matches_oracle(ucc_index, question, monkeypatch, caplog):
    from legal_rag_amd import encoders
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    from oracle import fusion as OF
    cfg, chunks = ucc_index
    ce = encoders.HashingCrossScorer()
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: ce)
    r = hr.HybridRetriever(cfg)
    assert r.colbert is not None and r.colbert.enabled
    with caplog.at_level(logging.INFO, logger="legalrag.retrieval.hybrid_retriever"):
        hits = r.search(question, top_k=10)
    assert any("[retrieval] dense=" in m and "rerank=" in m for m in caplog.messages)
    d, b, c = oracle_channels(cfg, chunks, question, 10)
    # per-channel parity: bm25 bit-exact, dense / colbert within 1e-4
    gb = r.search_bm25(question, 10)
    assert [(h.chunk.id, h.score) for h in gb] == b
    gd = r.search_dense(question, 10)
    assert [h.chunk.id for h in gd] == [i for i, _ in d]
    assert np.allclose([h.score for h in gd], [s for _, s in d], atol=1e-4)
    gc = r.search_colbert(question, 10)
    assert [h.chunk.id for h in gc] == [i for i, _ in c]
    assert np.allclose([h.score for h in gc], [s for _, s in c], atol=1e-4)
    # orchestration: run the oracle's search() on the GPU channel outputs (identical
    # inputs -> bit-exact fusion / rerank), with the CE scoring the same str(hit) text
    by_id = {x.id: x for x in chunks}
    fused_for_text = {h.chunk.id: h for h in r._fuse(dense_hits=r.search_dense(question, 10),
                                                     bm25_hits=r.search_bm25(question, 10),
                                                     colbert_hits=r.search_colbert(question, 10))}
    exp = OF.search([(h.chunk.id, h.score) for h in gd], [(h.chunk.id, h.score) for h in gb],
                    [(h.chunk.id, h.score) for h in gc], top_k=10, knobs={},
                    ce_score=lambda ids: ce.score_batch(question, [str(fused_for_text[i]) for i in ids]))
    assert_hits_equal_mod_ties([dump(h) for h in hits], exp)
    assert all(h.chunk is by_id[h.chunk.id] or h.chunk == by_id[h.chunk.id] for h in hits)


def test_search_batch_equals_single_queries(ucc_index):
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    batch = r.search_batch(QUESTIONS, top_k=10)
    for q, got in zip(QUESTIONS, batch):
        exp = r.search(q, top_k=10)
        assert [h.chunk.id for h in got] == [h.chunk.id for h in exp]
        # a batch of queries takes the 32-query-tile MFMA form of the dense scan, a single
        # query the GEMV form: the fp32 dot products are summed in a different order, so dense
        # scores (and everything derived from them) agree to rounding, not bit for bit
        assert np.allclose([h.score for h in got], [h.score for h in exp], rtol=0, atol=2e-5)
        for g, e in zip(got, exp):
            gb, eb = g.score_breakdown, e.score_breakdown
            assert gb["channel"] == eb["channel"] and gb["bm25_norm"] == eb["bm25_norm"]
            for key in ("rrf_norm", "weighted_sum", "dense_norm", "colbert_norm"):
                assert abs(gb[key] - eb[key]) <= 2e-5, key  # minmax divides by a small range
            for ch in ("dense", "bm25", "colbert"):
                assert abs(gb["channel_contrib"][ch] - eb["channel_contrib"][ch]) <= 2e-5


class IdReranker:
    """Cross-encoder stand-in whose score depends on (query, chunk id) only — the repr of a hit carries its fused
    score to the last digit, and a batch takes a different dense kernel form than a single query (rounding-level
    differences), so a text-hashing stand-in would jitter between the two paths."""

    def __init__(self):
        self.pair_calls = self.batch_calls = 0

    @staticmethod
    def _s(query, doc):
        import hashlib
        import re
        cid = re.search(r"LawChunk\(id='([^']+)'", doc).group(1)
        return int.from_bytes(hashlib.blake2b((query + "\0" + cid).encode(), digest_size=4).digest(), "little") / 2**32

    def score_batch(self, query, docs):
        self.batch_calls += 1
        return [self._s(query, d) for d in docs]

    def score_pairs(self, pairs):
        self.pair_calls += 1
        return [self._s(q, d) for q, d in pairs]


def test_search_batch_with_rerank_equals_single_queries(ucc_index, monkeypatch):
    """The reference's default configuration (ColBERT ON, rerank ON, config.py:97,119): search_batch(qs)[i] ==
    search(qs[i]) — the batch form runs the rerank stage too (cross-encoder fed over all queries' candidates in one
    score_pairs call, ONE blend launch), it no longer drops it (hybrid_retriever.py:324-356)."""
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    assert cfg2.retrieval.enable_rerank and cfg2.retrieval.enable_colbert
    ce = IdReranker()
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: ce)
    r = hr.HybridRetriever(cfg2)
    qs = QUESTIONS + ["", "zzzz qqqq", "Short Titles"]
    for top_k in (10, 4):
        ce.pair_calls = ce.batch_calls = 0
        batch = r.search_batch(qs, top_k=top_k)
        assert ce.pair_calls == 1 and ce.batch_calls == 0
        assert len(batch) == len(qs)
        for q, got in zip(qs, batch):
            exp = r.search(q, top_k=top_k)
            assert [h.chunk.id for h in got] == [h.chunk.id for h in exp], q
            assert [h.rank for h in got] == list(range(1, len(got) + 1))
            assert [h.source for h in got] == [h.source for h in exp]
            assert np.allclose([h.score for h in got], [h.score for h in exp], rtol=0, atol=2e-5)
            for g, e in zip(got, exp):
                assert g.score_breakdown.get("rerank_raw") == e.score_breakdown.get("rerank_raw")
                assert g.score_breakdown.get("rerank_beta") == e.score_breakdown.get("rerank_beta")
                if "rerank_norm" in e.score_breakdown:
                    assert g.score_breakdown["rerank_norm"] == e.score_breakdown["rerank_norm"]
        assert any(h.source == "rerank" for hits in batch for h in hits)
    # a reranker without score_pairs (duck-typed, the reference's protocol) is fed query by query
    class Plain:
        def score_batch(self, query, docs):
            return IdReranker().score_batch(query, docs)
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: Plain())
    again = r.search_batch(qs, top_k=10)
    first = r.search_batch(qs, top_k=10)
    assert [[(h.chunk.id, h.score) for h in a] for a in again] == [[(h.chunk.id, h.score) for h in a] for a in first]


def test_search_batch_runs_the_graph_stage_per_decision(ucc_index, monkeypatch):
    """decisions[i].mode == GRAPH_AUGMENTED cuts query i's fused list to the seeds and appends the graph channel's
    hits before rerank / dedup, exactly as search(decision=...) does (hybrid_retriever.py:312-322)."""
    import types
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    from legal_rag_amd.schemas import RetrievalHit
    cfg, chunks = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_graph = True
    cfg2.retrieval.graph_seed_k = 4
    ce = IdReranker()
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: ce)
    r = hr.HybridRetriever(cfg2)
    row = {c.id: i for i, c in enumerate(chunks)}

    def graph_search(question, seeds, decision=None, top_k=10, **kw):  # neighbours = the next chunk of every seed
        return [RetrievalHit(chunk=chunks[(row[h.chunk.id] + 1) % len(chunks)], score=0.9 - 0.1 * j, rank=j + 1,
                             source="graph", score_breakdown={"channel": ["graph"], "graph_depth": 1})
                for j, h in enumerate(seeds)]
    r.graph = types.SimpleNamespace(search=graph_search)
    dec = types.SimpleNamespace(mode="RoutingMode.GRAPH_AUGMENTED")
    plain = types.SimpleNamespace(mode="RoutingMode.RAG")
    qs = QUESTIONS[:4]
    decisions = [dec, None, plain, dec]
    batch = r.search_batch(qs, top_k=10, decisions=decisions)
    graph_seen = 0
    for q, d, got in zip(qs, decisions, batch):
        exp = r.search(q, top_k=10, decision=d)
        assert [h.chunk.id for h in got] == [h.chunk.id for h in exp], q
        assert np.allclose([h.score for h in got], [h.score for h in exp], rtol=0, atol=2e-5)
        assert [h.score_breakdown["channel"] for h in got] == [h.score_breakdown["channel"] for h in exp]
        graph_seen += sum("graph" in h.score_breakdown["channel"] for h in got)
        if d is dec:
            assert len(got) <= 8  # 4 seeds + their 4 neighbours, before dedup
    assert graph_seen > 0
    with pytest.raises(ValueError, match="one entry per question"):
        r.search_batch(qs, decisions=[dec])


def test_search_native_stage_equals_per_channel_path(ucc_index, monkeypatch):
    """search() keeps the whole retrieval stage in HBM (one stream, one synchronise) when every
    channel is the package's own retriever; AMDR_SEARCH_NATIVE=0 pins the per-channel path of the
    reference's structure.  Same kernels, same inputs: identical hits, scores and breakdowns —
    including the strings handed to the cross-encoder."""
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.rerank_ce_model = "hashing"  # explicit stand-in cross-encoder (rerankers.RerankerFactory)
    r = HybridRetriever(cfg2)
    assert r._native_channels(10) is not None
    for top_k in (3, 10, 25):
        for q in QUESTIONS + ["", "   ", "zzzz qqqq"]:
            monkeypatch.setenv("AMDR_SEARCH_NATIVE", "0")
            exp = [dump(h) for h in r.search(q, top_k=top_k)]
            monkeypatch.delenv("AMDR_SEARCH_NATIVE")
            got = [dump(h) for h in r.search(q, top_k=top_k)]
            assert got == exp, (q, top_k)


def test_search_one_launch_step_equals_the_separate_launches(ucc_index, monkeypatch):
    """search() without ColBERT issues ONE launch per query (amdr_hybrid_small_device: BM25 + dense + fusion of a
    serving corpus); AMDR_HYBRID_SMALL=0 pins bm25.search_device + dense.search_fuse_device.  Identical hits, scores
    and breakdowns through the API, and both equal to the per-channel path (AMDR_SEARCH_NATIVE=0)."""
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_colbert = False
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    assert r._native_channels(10) is not None
    for top_k in (1, 10, 16):
        for q in QUESTIONS + ["", "the", "zzzz qqqq"]:
            monkeypatch.setenv("AMDR_HYBRID_SMALL", "0")
            exp = [dump(h) for h in r.search(q, top_k=top_k)]
            monkeypatch.setenv("AMDR_HYBRID_SMALL", "1")
            got = [dump(h) for h in r.search(q, top_k=top_k)]
            monkeypatch.delenv("AMDR_HYBRID_SMALL")
            monkeypatch.setenv("AMDR_SEARCH_NATIVE", "0")
            per = [dump(h) for h in r.search(q, top_k=top_k)]
            monkeypatch.delenv("AMDR_SEARCH_NATIVE")
            assert got == exp == per, (q, top_k)


def test_failing_colbert_encoder_degrades_locally_but_raises_on_a_sharded_index(ucc_index, monkeypatch):
    """A ColBERT query encoder that raises empties the channel (the reference swallows ColBERT errors,
    hybrid_retriever.py:244-245) — except on a row-sharded index, where dropping a channel would be a rank-local decision
    inside an SPMD exchange (the other ranks all-gather three packed channels): there the error propagates."""
    import types
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    store, _bm, col = r._native_channels(10)
    assert col is not None

    def boom(*a, **k):
        raise RuntimeError("encoder down")
    for name in ("encode_queries_tensor", "encode_queries", "encode_query"):
        if hasattr(col._encoder, name):
            monkeypatch.setattr(col._encoder, name, boom)
    hits = r.search(QUESTIONS[0], top_k=5)
    assert hits and all("colbert" not in (h.score_breakdown or {}).get("channel", []) for h in hits)
    monkeypatch.setattr(store.index, "spec", types.SimpleNamespace(group=None), raising=False)
    with pytest.raises(RuntimeError, match="encoder down"):
        r.search(QUESTIONS[0], top_k=5)


def test_search_and_search_batch_from_two_threads(ucc_index):
    """Service threads in search() while another thread drives search_batch() on the same
    singletons (VectorStore / BM25 / MaxSim handles): every result equals the single-threaded one."""
    import threading
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    key = lambda hits: [(h.chunk.id, round(h.score, 4)) for h in hits]  # noqa: E731
    single = {q: key(r.search(q, top_k=10)) for q in QUESTIONS}
    batch_q = QUESTIONS * 7
    batch_exp = [key(h) for h in r.search_batch(batch_q, top_k=10)]
    errors = []

    def singles():
        for _ in range(40):
            for q in QUESTIONS:
                if key(r.search(q, top_k=10)) != single[q]:
                    errors.append(("search", q))

    def batches():
        for _ in range(25):
            if [key(h) for h in r.search_batch(batch_q, top_k=10)] != batch_exp:
                errors.append(("search_batch",))

    def host_api():  # the host-pointer entry points of the same handles, from yet another thread
        for _ in range(40):
            for q in QUESTIONS:
                if key(r.search_dense(q, 10))[:3] != key(r.search_dense(q, 10))[:3]:
                    errors.append(("dense", q))

    ts = [threading.Thread(target=f) for f in (singles, singles, batches, host_api)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_search_batch_arrays_equals_search_batch(ucc_index):
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    qs = QUESTIONS * 3
    hits = r.search_batch(qs, top_k=10)
    col = r.search_batch_arrays(qs, top_k=10)
    assert col["rows"].shape == (len(qs), 10) and col["zh_exact"].all()
    for q, hs in enumerate(hits):
        n = int(col["count"][q])
        assert n == len(hs)
        assert [col["chunks"][int(i)].id for i in col["rows"][q, :n]] == [h.chunk.id for h in hs]
        assert col["scores"][q, :n].tolist() == [h.score for h in hs]
        assert (col["rows"][q, n:] == -1).all()
        for j, h in enumerate(hs):
            assert col["values"][q, j, col["value_names"]["rrf_norm"]] == h.score_breakdown["rrf_norm"]
    with pytest.raises(ValueError):
        r.search_batch_arrays(["ok", "  "], top_k=5)
    # values=False: the lean columnar form, compacted on the device (amdr_fuse_compact_device) — the same rows / scores /
    # masks / counts as the full form's first top_k columns
    for k in (3, 10, 50):
        full = r.search_batch_arrays(QUESTIONS * 5, top_k=k)
        lean = r.search_batch_arrays(QUESTIONS * 5, top_k=k, values=False)
        w = lean["rows"].shape[1]
        assert "values" not in lean and w == min(k, full["rows"].shape[1])
        for key in ("rows", "scores", "channel_mask"):
            assert np.array_equal(lean[key], full[key][:, :w]), (k, key)
        assert np.array_equal(lean["count"], full["count"])


def test_batch_tokeniser_and_caller_supplied_embeddings(ucc_index, monkeypatch):
    """search_batch_arrays tokenises the whole batch in one native call (BM25Retriever.term_ids_batch) and accepts the
    caller's own query embeddings: both must give exactly what the per-query Python path gives."""
    from legal_rag_amd import text
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    qs = QUESTIONS * 3 + ["What is § 2-314?", "rate of 3.5% p.a. (a) C++", "x"]
    terms, q_ptr, exact = r.bm25.term_ids_batch(qs)
    assert exact.all() and q_ptr[-1] == len(terms)
    for i, q in enumerate(qs):
        assert terms[q_ptr[i]:q_ptr[i + 1]].tolist() == r.bm25.bm25.term_ids(text.jieba_cut(q))
    base = r.search_batch_arrays(qs, top_k=10)
    emb = r.dense.store._embed(qs, is_query=True)
    for q_emb in (emb, __import__("torch").from_numpy(emb).cuda()):
        got = r.search_batch_arrays(qs, top_k=10, q_emb=q_emb)
        assert np.array_equal(got["rows"], base["rows"]) and np.array_equal(got["scores"], base["scores"])
    with pytest.raises(ValueError, match="q_emb must be"):
        r.search_batch_arrays(qs, top_k=10, q_emb=emb[:3])
    # with a registered segmenter every query goes through it (the native rule is only jieba's no-dictionary case)
    calls = []
    monkeypatch.setattr(text, "_custom_cut", lambda s_: calls.append(s_) or text.jieba_cut_restated(s_))
    t2, p2, _ = r.bm25.term_ids_batch(qs[:4])
    assert len(calls) == 4 and np.array_equal(t2, terms[: q_ptr[4]]) and np.array_equal(p2, q_ptr[:5])
    # with jieba installed (every deployment of the reference) the batch STAYS native: only a query that holds one of
    # the ASCII entries of jieba's dictionary goes to jieba itself — per query, not per process
    monkeypatch.setattr(text, "_custom_cut", None)
    seen = []

    class FakeJieba:
        @staticmethod
        def cut(s_):
            seen.append(s_)
            return text.jieba_cut_restated(s_)
    monkeypatch.setattr(text, "HAVE_JIEBA", True)
    monkeypatch.setattr(text, "_jieba", FakeJieba)
    t3, p3, e3 = r.bm25.term_ids_batch(qs)
    assert seen == ["rate of 3.5% p.a. (a) C++"] and e3.all()
    assert np.array_equal(t3, terms) and np.array_equal(p3, q_ptr)


def test_error_conventions(tmp_path):
    """Missing dense files -> FileNotFoundError; missing bm25 -> RuntimeError;
    missing colbert meta is swallowed at construction."""
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    cfg.retrieval.encoder_backend = "hashing"
    r = HybridRetriever(cfg)
    assert r.colbert is None
    with pytest.raises(FileNotFoundError):
        r.search_dense("x", 3)
    with pytest.raises(RuntimeError):
        r.search_bm25("x", 3)
    assert r.search_colbert("x", 3) == [] and r.search_graph("x") == []


def test_reference_built_style_artifacts_load(tmp_path):
    """A bm25.pkl naming rank_bm25.BM25Okapi and a faiss.index in FAISS's container
    format load and search (artifact compatibility)."""
    from legal_rag_amd import artifacts
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.bm25_retriever import BM25Retriever
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")[:50]
    build_bm25_index(cfg, chunks)
    assert b"rank_bm25" in open(cfg.retrieval.bm25_index_file, "rb").read()
    br = BM25Retriever(cfg)
    got = br.search("seller goods", 5)
    from oracle import bm25 as OB
    ob = OB.BM25Okapi([OB.tokenize_en(c.text) for c in chunks])
    assert [(c.id, s) for c, s in got] == [(chunks[i].id, s) for i, s in OB.search(ob, ["seller", " ", "goods"], 5)]


def test_incremental_add_and_hot_reload(tmp_path):
    """§8f-3: IncrementalDenseBuilder / IncrementalBM25Builder semantics — id-dedup, append,
    persisted artifacts, searches see the new rows, a second retriever reloads by mtime."""
    import json
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.bm25_retriever import BM25Retriever
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index
    from legal_rag_amd.retrieval.builders.incremental_bm25_builder import IncrementalBM25Builder
    from legal_rag_amd.retrieval.builders.incremental_dense_builder import IncrementalDenseBuilder
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    from legal_rag_amd.retrieval.dense_retriever import DenseRetriever
    from legal_rag_amd import artifacts
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    cfg.retrieval.encoder_backend = "hashing"
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")
    base, extra = chunks[:60], chunks[55:90]  # 5 overlapping ids
    build_faiss_index(cfg, base)
    build_bm25_index(cfg, base)
    dr = DenseRetriever(cfg)
    target = extra[-1]
    assert all(h.chunk.id != target.id for h in dr.search(target.text[:200], 5))
    inc = tmp_path / "incoming.jsonl"
    inc.write_text("".join(json.dumps(c.model_dump(), ensure_ascii=False) + "\n" for c in extra), encoding="utf-8")
    assert IncrementalDenseBuilder(cfg).add_jsonl(inc) == 30
    assert IncrementalDenseBuilder(cfg).add_jsonl(inc) == 0          # idempotent
    X, _ = artifacts.read_faiss_index(cfg.retrieval.faiss_index_file)
    meta = artifacts.read_faiss_meta(cfg.retrieval.faiss_meta_file)
    assert X.shape[0] == len(meta) == 90 and [c.id for c in meta] == [c.id for c in chunks[:90]]
    assert dr.search(target.text[:200], 3)[0].chunk.id == target.id  # same store object sees the new rows
    br = BM25Retriever(cfg)
    assert len(br.search("goods", 100)) == 60
    assert IncrementalBM25Builder(cfg).add_jsonl(inc) == 30
    assert len(br.search("goods", 100)) == 90                          # mtime-triggered reload
    with pytest.raises(FileNotFoundError):
        IncrementalDenseBuilder(cfg).add_jsonl(tmp_path / "missing.jsonl")


def test_concurrent_searches_on_shared_singletons(ucc_index):
    """The /retrieve service runs RETRIEVER.search from a thread pool on shared
    singletons: results must not depend on interleaving."""
    from concurrent.futures import ThreadPoolExecutor
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, _ = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    expect = {q: [(h.chunk.id, h.score) for h in r.search(q, top_k=10)] for q in QUESTIONS}
    work = QUESTIONS * 12
    with ThreadPoolExecutor(max_workers=8) as ex:
        got = list(ex.map(lambda q: (q, [(h.chunk.id, h.score) for h in r.search(q, top_k=10)]), work))
    for q, res in got:
        assert res == expect[q]


def test_colbert_retriever_serves_a_plaid_layout_index(tmp_path):
    """§8 a-12 / f-1: a ColBERT index directory in colbert-ai's on-disk layout (centroids + packed
    4-bit residual codes) is decompressed at load and searched by the MaxSim kernel; scores equal
    the oracle's MaxSim over the decompressed token embeddings."""
    from test_artifacts import _write_plaid_fixture
    from legal_rag_amd import artifacts
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index
    from legal_rag_amd.retrieval.colbert_retriever import ColBERTRetriever
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    from oracle import maxsim as OM
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    cfg.retrieval.encoder_backend = "hashing"
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")[:40]
    out_dir = build_colbert_index(cfg, chunks)                       # writes colbert_meta.jsonl + the fp32 store
    D, doc_ptr = artifacts.read_token_store(out_dir)
    (out_dir / "amdr_tokens.npz").unlink()                            # keep only what colbert-ai would have written
    exp_tokens = _write_plaid_fixture(out_dir, D, np.diff(doc_ptr).tolist(), nbits=4, n_centroids=64, chunk_docs=16)
    r = ColBERTRetriever(cfg)
    q = "what warranty does a merchant give that goods are merchantable"
    got = r.search(q, top_k=10)
    assert len(got) == 10
    qt = r._encoder.encode_query(q)[None]
    es, ei = OM.maxsim_topk(qt, exp_tokens, doc_ptr, 10)
    assert [c.id for c, _ in got] == [chunks[i].id for i in ei[0]]
    assert np.allclose([s for _, s in got], es[0], atol=1e-4)


def test_depth_beyond_kernel_limit_is_handled_uniformly(ucc_index):
    """AMDR_MAX_K = 256: every per-channel search accepts a deeper request (full scores on the device
    + stable host sort), results continue the 256-deep ones; the fused search raises ONE clear error."""
    from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever
    cfg, chunks = ucc_index
    cfg2 = copy.deepcopy(cfg)
    cfg2.retrieval.enable_rerank = False
    r = HybridRetriever(cfg2)
    q = QUESTIONS[0]
    n = len(chunks)
    for fn in (r.search_dense, r.search_bm25, r.search_colbert):
        deep, ref = fn(q, 300), fn(q, 256)
        assert len(deep) == min(300, n) and len(ref) == min(256, n)
        assert [h.chunk.id for h in deep[:len(ref)]] == [h.chunk.id for h in ref]
        assert np.allclose([h.score for h in deep[:len(ref)]], [h.score for h in ref], atol=1e-5)
        assert all(a.score >= b.score for a, b in zip(deep, deep[1:]))
    with pytest.raises(ValueError, match="exceeds the fusion kernel"):
        r.search(q, top_k=300)
    for batch_form in (r.search_batch, r.search_batch_arrays):  # the batch forms refuse the same way (no silent clamp)
        with pytest.raises(ValueError, match="exceeds the fusion kernel"):
            batch_form([q], top_k=300)
    big = [h for h in r.search_dense(q, 200)] * 2
    with pytest.raises(ValueError, match="exceed the fusion kernel"):
        r._fuse(dense_hits=big, bm25_hits=[], colbert_hits=[])


def test_build_index_and_evaluate_retrieval_clis(tmp_path, capsys):
    """BASELINE configs[0]'s recipe through the two CLIs (scripts/build_index.py:66-119 ->
    scripts/evaluate_retrieval.py:65-125 in the reference): processed jsonl -> indexes in the reference's artifact
    layout (versioned + activated) -> the evaluation summary table, on the GPU engine."""
    import shutil
    import sys
    sys.path.insert(0, str(GOLDEN.parent.parent / "scripts"))
    import build_index
    import evaluate_retrieval
    from legal_rag_amd.retrieval.vector_store import VectorStore
    data = tmp_path / "data"
    (data / "processed").mkdir(parents=True)
    lines = (GOLDEN / "corpus" / "law_en.jsonl").read_text(encoding="utf-8").splitlines()[:150]
    (data / "processed" / "law_en.jsonl").write_text("\n".join(lines) + "\n", encoding="utf-8")
    build_index.main(["--data-dir", str(data), "--encoder-backend", "hashing", "--index-version", "v1", "--activate"])
    root = data / "index" / "en"
    assert (root / "ACTIVE").read_text() == "v1"
    for rel in ("faiss/faiss.index", "faiss/faiss_meta.jsonl", "bm25.pkl", "colbert/colbert_meta.jsonl"):
        assert (root / "versions" / "v1" / rel).exists(), rel
    out = tmp_path / "rows.jsonl"
    VectorStore._instances_by_key.clear()
    evaluate_retrieval.main(["--data-dir", str(data), "--lang", "en", "--synthetic", "--limit", "20",
                             "--encoder-backend", "hashing", "--no-rerank", "--output", str(out)])
    txt = capsys.readouterr().out
    assert "Evaluation Summary over 20 queries" in txt
    rows = [__import__("json").loads(l) for l in out.read_text().splitlines()]
    systems = {r["system"] for r in rows}
    assert systems == {"bm25", "dense", "colbert", "fused", "fused+graph", "hybrid"} and len(rows) == 6 * 20
    mean = lambda name, k: float(np.mean([r[k] for r in rows if r["system"] == name]))  # noqa: E731
    assert mean("fused", "R@10") >= max(mean("bm25", "R@10"), 0.5)  # fusion is not worse than the weakest channel
    assert mean("hybrid", "R@10") == mean("fused", "R@10") or mean("hybrid", "R@10") > 0.5
    assert all(0.0 <= r[k] <= 1.0 for r in rows for k in ("R@5", "R@10", "MRR@10", "nDCG@10", "Hit@3", "Hit@10"))
    shutil.rmtree(data)
