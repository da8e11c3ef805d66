"""BASELINE configs[3] as far as one GPU goes: the FULL Civil-Code-zh (1 260 chunks) and UCC-en
(591 chunks) fixtures, the reference's default hybrid (ColBERT ON, rerank ON: config.py:97,119):
dense + BM25 + ColBERT MaxSim -> fuse -> min_final filter -> rerank blend, batched on the device
(the bench's `full_hybrid_rerank` step) and through the language-routed Python API — both against
the CPU oracle on identical inputs.  jieba is absent offline, so zh BM25 runs on the explicitly
chosen one-character stand-in and says so (zh_exact False); the multi-GPU part of configs[3] is
covered by tests/test_sharding_gloo_gpu.py and tests/test_multilang_shard_gpu.py."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lang,n_chunks", [("en", 591), ("zh", 1260)])
def test_batched_full_hybrid_with_rerank_equals_oracle(lang, n_chunks):
    import torch

    import bench
    from legal_rag_amd import _native
    W = bench.build_corpus(lang, colbert=True)
    assert len(W["chunks"]) == n_chunks
    R = bench.Resident(torch, W, 0, rep=1, colbert=True)
    K, beta, top_n = 10, 0.35, 30
    R.reserve(K)
    params = _native.make_fuse_params(w_dense=0.6, w_bm25=0.4, w_colbert=0.35, min_final_score=0.2)
    dev = torch.device("cuda", 0)
    ce = torch.sigmoid(4.0 * (R.q_emb.double() @ torch.from_numpy(W["X"]).to(dev).double().T))
    ce = (ce + 1e-6 * torch.rand(ce.shape, generator=torch.Generator(device=dev).manual_seed(3), device=dev,
                                 dtype=torch.float64)).contiguous()
    res = R.search_batch(params, K)
    ce_raw = torch.gather(ce, 1, res.ids[:, :top_n].clamp(min=0)).contiguous()
    res = R.eng.rerank_blend(res, ce_raw, beta)
    torch.cuda.synchronize()
    ids, cnt = res.ids[:, :K].cpu().numpy(), res.count.cpu().numpy()
    sample = list(range(0, R.nq0, max(1, R.nq0 // 60)))
    exp = bench.oracle_pipeline(W, sample, K, ce=ce.cpu().numpy(), beta=beta, top_n=top_n)
    for j, qi in enumerate(sample):
        got = [int(x) for x in ids[qi, :min(int(cnt[qi]), K)]]
        assert got == exp[j], (lang, qi)
    assert W["zh_exact"] == (lang == "en" or __import__("legal_rag_amd").text.zh_exact())
    R.close()


def test_side_stream_channels_equal_the_one_stream_batch_eager_and_captured(monkeypatch):
    """engine.search_batch with ColBERT: the dense and BM25 channels on a side stream beside MaxSim (default) against all
    three on the caller's stream (AMDR_ENGINE_OVERLAP=0) — same bits; and the forked form captured into a hipGraph and
    replayed on fresh inputs."""
    import torch

    import bench
    from legal_rag_amd import _native
    W = bench.build_corpus("en", colbert=True)
    R = bench.Resident(torch, W, 0, rep=1, colbert=True)
    K = 10
    R.reserve(K)
    params = _native.make_fuse_params(w_dense=0.6, w_bm25=0.4, w_colbert=0.35, min_final_score=0.2)

    def snap():
        res = R.search_batch(params, K)
        torch.cuda.synchronize()
        return res, (res.ids.cpu().numpy().copy(), res.vals.cpu().numpy().copy(), res.count.cpu().numpy().copy())

    monkeypatch.setenv("AMDR_ENGINE_OVERLAP", "0")
    _, one = snap()
    monkeypatch.setenv("AMDR_ENGINE_OVERLAP", "1")
    for _ in range(3):
        _, two = snap()
        assert (one[2] == two[2]).all()
        for q in range(one[0].shape[0]):
            c = int(one[2][q])
            assert (one[0][q, :c] == two[0][q, :c]).all() and (one[1][q, :c].view("uint64") == two[1][q, :c].view("uint64")).all(), q  # (vals: f64 [.., 9])
    assert R.eng.__dict__.get("_side_stream") is not None
    # captured: the fork / join become edges of the graph; replay after the inputs changed and were restored
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        R.search_batch(params, K)
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        res = R.search_batch(params, K)
    keep = R.q_emb.clone()
    R.q_emb.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert not (res.ids.cpu().numpy()[:, :K] == one[0][:, :K]).all()  # (the dense channel saw zero queries)
    R.q_emb.copy_(keep)
    g.replay()
    torch.cuda.synchronize()
    ids, cnt = res.ids.cpu().numpy(), res.count.cpu().numpy()
    assert (cnt == one[2]).all()
    for q in range(ids.shape[0]):
        assert (ids[q, :int(cnt[q])] == one[0][q, :int(cnt[q])]).all(), q
    R.close()


def test_language_routed_api_default_hybrid_on_full_fixtures(tmp_path):
    """ByLangRetriever -> HybridRetriever(lang).search with ColBERT and rerank ON over the full
    fixtures, indexes built by the product builders; a sample of queries per language against the
    oracle's search() fed with the oracle's own channel results."""
    from legal_rag_amd import encoders, text
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.evaluation import synthetic_queries
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index
    from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index
    from legal_rag_amd.retrieval.by_lang_retriever import ByLangRetriever
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    from oracle import bm25 as OB
    from oracle import dense as OD
    from oracle import fusion as OF
    from oracle import maxsim as OM
    from conftest import GOLDEN
    from helpers import assert_hits_equal_mod_ties
    base = AppConfig.for_data_dir(str(tmp_path), "zh")
    base.retrieval.encoder_backend = "hashing"
    base.retrieval.rerank_ce_model = "hashing"
    base.retrieval.enable_graph = False
    base.retrieval.zh_tokenizer = "char"  # jieba absent: explicit opt-in, reported as zh_exact False
    assert base.retrieval.enable_colbert and base.retrieval.enable_rerank  # the reference's defaults
    corp = {}
    for lang in ("zh", "en"):
        cfg = base.with_lang(lang)
        chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), f"law_{lang}.jsonl")
        build_faiss_index(cfg, chunks)
        build_bm25_index(cfg, chunks)
        build_colbert_index(cfg, chunks)
        corp[lang] = chunks
    r = ByLangRetriever(base)
    ce = encoders.HashingCrossScorer()
    emb, te = encoders.HashingEmbedder(768), encoders.HashingTokenEmbedder()
    for lang in ("zh", "en"):
        chunks = corp[lang]
        X = emb.encode([c.text for c in chunks])
        mode = "char" if lang == "zh" else None
        ob = OB.BM25Okapi([OB.tokenize_en(c.text) if lang == "en" else text.jieba_cut(c.text, mode) for c in chunks])
        mats = [te.encode_doc(c.text.strip()) for c in chunks]
        ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])])
        Dtok = np.concatenate(mats)
        qs = [q for q, _, _ in synthetic_queries(chunks, seed=0)]
        for q in qs[:: max(1, len(qs) // 6)][:6]:
            hits = r.search(q, top_k=10)
            assert hits and all(h.chunk.lang == lang for h in hits) and all(h.source == "rerank" for h in hits)
            ds, di = OD.flatip_topk(X, emb.encode_queries([q]), 10)
            b = OB.search(ob, text.jieba_cut(q, mode), 10)
            cs, ci = OM.maxsim_topk(te.encode_query(q.strip())[None], Dtok, ptr, 10)
            d = [(chunks[i].id, float(s)) for s, i in zip(ds[0], di[0]) if i >= 0]
            c = [(chunks[i].id, float(s)) for s, i in zip(cs[0], ci[0]) if i >= 0]
            hr_ = r._retrievers[lang]
            # the cross-encoder scores the str() of the fused hit (rerankers.py:78-86): take the product's own
            # fused hits for the text, the oracle for the arithmetic
            fused_txt = {h.chunk.id: h for h in hr_._fuse(dense_hits=hr_.search_dense(q, 10),
                                                          bm25_hits=hr_.search_bm25(q, 10),
                                                          colbert_hits=hr_.search_colbert(q, 10))}
            gd, gb, gc = hr_.search_dense(q, 10), hr_.search_bm25(q, 10), hr_.search_colbert(q, 10)
            assert [h.chunk.id for h in gb] == [chunks[i].id for i, _ in b] and [h.score for h in gb] == [s for _, s in b]
            assert [h.chunk.id for h in gd] == [i for i, _ in d] and [h.chunk.id for h in gc] == [i for i, _ in c]
            exp = OF.search([(h.chunk.id, h.score) for h in gd], [(h.chunk.id, h.score) for h in gb],
                            [(h.chunk.id, h.score) for h in gc], top_k=10, knobs={},
                            ce_score=lambda ids: ce.score_batch(q, [str(fused_txt[i]) for i in ids]))
            got = [{"id": h.chunk.id, "score": float(h.score), "rank": h.rank, "source": h.source,
                    "breakdown": {k: v for k, v in h.score_breakdown.items() if k != "zh_exact"}} for h in hits]
            assert_hits_equal_mod_ties(got, exp)
            if lang == "zh" and not text.zh_exact():
                assert all(h.score_breakdown.get("zh_exact") is False for h in hits)
            else:
                assert all("zh_exact" not in h.score_breakdown for h in hits)
