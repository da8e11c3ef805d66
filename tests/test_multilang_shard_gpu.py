"""BASELINE configs[3] shape on one GPU: zh + en corpora behind ByLangRetriever with
rerank, and the row-sharded layout (W shards in one process: local top-k per shard with
GLOBAL BM25 statistics -> merge kernel -> fusion) equal to the unsharded engine."""
import copy

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def two_lang_index(tmp_path_factory):
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    data = tmp_path_factory.mktemp("data2")
    base = AppConfig.for_data_dir(str(data), "zh")
    base.retrieval.encoder_backend = "hashing"
    base.retrieval.enable_colbert = False
    base.retrieval.zh_tokenizer = "char"  # jieba is absent here: explicit opt-in to the stand-in (text.py)
    out = {}
    for lang in ("zh", "en"):
        cfg = base.with_lang(lang)
        chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), f"law_{lang}.jsonl")
        build_faiss_index(cfg, chunks)
        build_bm25_index(cfg, chunks)
        out[lang] = chunks
    return base, out


def test_by_lang_routing_with_rerank(two_lang_index, monkeypatch):
    from legal_rag_amd import encoders, text
    from legal_rag_amd.retrieval import hybrid_retriever as hr
    from legal_rag_amd.retrieval.by_lang_retriever import ByLangRetriever
    from oracle import bm25 as OB
    base, chunks = two_lang_index
    ce = encoders.HashingCrossScorer()
    monkeypatch.setattr(hr.RerankerFactory, "create", lambda self, top_k: ce)
    r = ByLangRetriever(base)
    zh_q = "当事人订立合同可以采用书面形式吗"
    en_q = "when does a security interest attach to collateral"
    zh_hits = r.search(zh_q, top_k=5)
    en_hits = r.search(en_q, top_k=5)
    assert zh_hits and all(h.chunk.lang == "zh" for h in zh_hits)
    assert en_hits and all(h.chunk.lang == "en" for h in en_hits)
    assert set(r._retrievers) == {"zh", "en"}
    assert all(h.source == "rerank" and "rerank_norm" in h.score_breakdown for h in zh_hits + en_hits)
    # zh BM25 channel vs the oracle on the same (fallback, per-character) tokenisation
    toks = [text.jieba_cut(c.text, "char") for c in chunks["zh"]]
    ob = OB.BM25Okapi(toks)
    got = r._retrievers["zh"].search_bm25(zh_q, 10)
    exp = OB.search(ob, text.jieba_cut(zh_q, "char"), 10)
    assert [(h.chunk.id, h.score) for h in got] == [(chunks["zh"][i].id, s) for i, s in exp]
    if not text.zh_exact():  # the stand-in never goes unmarked
        assert r._retrievers["zh"].bm25.index_tokenizer == "char"
        assert all(h.score_breakdown.get("zh_exact") is False for h in got)
        assert all(h.score_breakdown.get("zh_exact") is False for h in zh_hits)
        assert all("zh_exact" not in h.score_breakdown for h in en_hits)


def test_row_sharded_engine_equals_unsharded(two_lang_index):
    import torch
    from legal_rag_amd import _native, encoders, text
    from legal_rag_amd.bm25_model import BM25Okapi
    from legal_rag_amd.retrieval import sharding
    from legal_rag_amd.retrieval.engine import HybridEngine
    _, chunks = two_lang_index
    ch = chunks["en"]
    n, K, W = len(ch), 10, 4
    emb = encoders.HashingEmbedder(768)
    X = emb.encode([c.text for c in ch])
    bm = BM25Okapi([text.tokenize_en(c.text) for c in ch])
    queries = [c.text[:80] for c in ch[::23]]
    Q = torch.from_numpy(emb.encode_queries(queries)).cuda()
    qt, qp = _native.BM25Index.pack_queries([bm.term_ids(text.jieba_cut(q)) for q in queries])
    qt, qp = torch.from_numpy(qt).cuda(), torch.from_numpy(qp).cuda()
    params = _native.make_fuse_params(min_final_score=0.2)
    full = HybridEngine(_native.DenseIndex(X), bm.gpu(0), None).search_batch(params, K, q_emb=Q, q_terms=qt, q_ptr=qp)
    torch.cuda.synchronize()
    exp = (full.ids.clone(), full.vals.clone(), full.count.clone(), full.bm25_scores.clone(), full.bm25_ids.clone())
    tp, pd, pt, idf, dl = bm.to_csr()
    term_of = np.repeat(np.arange(len(tp) - 1), np.diff(tp))
    parts = []
    for lo, hi in sharding.shard_bounds(n, W):
        keep = (pd >= lo) & (pd < hi)
        cnt = np.zeros(len(tp), dtype=np.int64)
        np.add.at(cnt, term_of[keep] + 1, 1)
        bmi = _native.BM25Index(np.cumsum(cnt), pd[keep] - lo, pt[keep], idf, dl[lo:hi], float(bm.avgdl), bm.k1, bm.b)
        eng = HybridEngine(_native.DenseIndex(X[lo:hi]), bmi, None)
        ds, di = eng.dense_topk(Q, K)
        bs, bi = eng.bm25_topk(qt, qp, K)
        torch.cuda.synchronize()
        parts.append((ds.clone(), sharding.to_global(di, lo).clone(), bs.clone(), sharding.to_global(bi, lo).clone(), eng))
    mds, mdi = sharding.native_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), K)
    mbs, mbi = sharding.native_merge(torch.stack([p[2] for p in parts]), torch.stack([p[3] for p in parts]), K)
    res = parts[0][4].fuse(params, len(queries), (mds, mdi), (mbs, mbi), None)
    torch.cuda.synchronize()
    assert torch.equal(mbi, exp[4]) and torch.equal(mbs, exp[3])       # BM25: bit-exact across the shard split
    assert torch.equal(res.ids, exp[0]) and torch.equal(res.count, exp[2])
    assert torch.allclose(res.vals, exp[1], atol=2e-5, rtol=0)          # dense: GEMV vs tile summation order
