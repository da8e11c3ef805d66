"""BASELINE configs[3] in its stated form, as far as one card allows: Civil-Code-zh + UCC-en, the reference's
default hybrid (ColBERT ON, rerank ON: config.py:97,119), ROW-SHARDED over two ranks.  Two `gloo` ranks share
device 0 (RCCL refuses two ranks on one card; on a multi-GPU node the same code runs on backend nccl = RCCL).
Every rank holds its row block of ALL THREE channels (chunk rows, doc-partitioned postings with global idf / avgdl,
the token vectors of its documents), runs the real kernels on it, the three per-shard top-k lists cross in ONE
all_gather_into_tensor, merge_parts_kernel x 3, then fusion, the candidates' cross-encoder scores and the rerank
blend are replicated — and every rank must end with the results of the UNSHARDED corpus:

  * engine level (`bench.run_full_hybrid_rerank`, the `full_hybrid_rerank_sharded` object of bench.py) against
    the CPU oracle of the unsharded corpus, both languages in full;
  * API level: `cfg.retrieval.shard = "rows"` behind ByLangRetriever / HybridRetriever.search, search_batch and the
    per-channel searches, against the unsharded product (itself pinned to the oracle by test_full_hybrid_gpu.py) and,
    per channel, against the oracle — on a corpus with a pair of IDENTICAL documents on either side of the shard
    boundary (exact ties in every channel, MaxSim included: the lower global id must win on every rank)."""
import json
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))

pytestmark = pytest.mark.gpu

TIE_LO = 5  # en chunk whose text is repeated at row n//2 + 3 (the first rows of rank 1's block)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return dist


def _engine_worker(rank, world, port, out_dir):
    dist = _init(rank, world, port)
    import bench
    out = bench.run_full_hybrid_rerank(torch, 0, 10, 2, dist, world, rank)
    if rank == 0:
        Path(out_dir, "engine.json").write_text(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def test_engine_three_channels_and_rerank_through_the_exchange(tmp_path):
    mp.spawn(_engine_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    out = json.loads((tmp_path / "engine.json").read_text())
    assert "sharded over 2 GPUs" in out["workload"]
    for lang, n in (("zh", 1260), ("en", 591)):
        info = out["per_lang"][lang]
        assert info["chunks"] == n and info["rows_this_rank"] == [0, n // 2]
        assert info["agreement_at_10_vs_oracle"] == 1.0, (lang, info)  # the oracle of the UNSHARDED corpus
        assert info["identical_on_every_rank"] is True
        assert info["recall_at_10"] > 0.85
    assert len(out["maxsim_ms_per_rank"]) == 2 and all(m > 0 for m in out["maxsim_ms_per_rank"])
    assert all(m > 0 for m in out["collective_plus_merge_ms_per_rank"])
    assert out["exchange_bytes_per_rank_per_step"] == out["queries_per_step"] * 3 * 10 * 16


# ------------------------------------------------------------------------------------------------ API level
def _base_cfg(data_dir, shard):
    from legal_rag_amd.config import AppConfig
    base = AppConfig.for_data_dir(str(data_dir), "zh")
    base.retrieval.encoder_backend = "hashing"
    base.retrieval.rerank_ce_model = "hashing"
    base.retrieval.enable_graph = False
    base.retrieval.zh_tokenizer = "char"  # jieba absent: explicit opt-in, reported as zh_exact False
    base.retrieval.shard = shard
    assert base.retrieval.enable_colbert and base.retrieval.enable_rerank  # the reference's defaults
    return base


def _corpora():
    from conftest import GOLDEN
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    out = {}
    for lang in ("zh", "en"):
        chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), f"law_{lang}.jsonl")
        if lang == "en":  # an identical document on the far side of the two-rank shard boundary
            hi = len(chunks) // 2 + 3
            chunks[hi] = chunks[hi].model_copy(update={"text": chunks[TIE_LO].text})
        out[lang] = chunks
    return out


def _queries(corp):
    from legal_rag_amd.evaluation import synthetic_queries
    qs = {}
    for lang, chunks in corp.items():
        all_q = [q for q, _, _ in synthetic_queries(chunks, seed=0)]
        qs[lang] = all_q[:: max(1, len(all_q) // 7)][:7]
    words = corp["en"][TIE_LO].text.split()
    qs["en"].append(" ".join(words[3:17]))  # a span of the duplicated document: the tied pair leads every channel
    return qs


def _dump(h):
    return {"id": h.chunk.id, "score": float(h.score), "rank": h.rank, "source": h.source,
            "breakdown": h.score_breakdown}


def _run_api(base, corp):
    """Everything the comparison looks at, through the public API only."""
    from legal_rag_amd.retrieval.by_lang_retriever import ByLangRetriever
    r = ByLangRetriever(base)
    qs = _queries(corp)
    out = {}
    for lang in ("zh", "en"):
        single = [[_dump(h) for h in r.search(q, top_k=10)] for q in qs[lang]]
        hr_ = r._retrievers[lang]
        batch = [[_dump(h) for h in hits] for hits in hr_.search_batch(qs[lang], top_k=10)]
        q = qs[lang][-1]
        chan = {name: [(h.chunk.id, float(h.score)) for h in fn(q, 10)]
                for name, fn in (("dense", hr_.search_dense), ("bm25", hr_.search_bm25), ("colbert", hr_.search_colbert))}
        out[lang] = {"single": single, "batch": batch, "channels": chan}
    return out


def _api_worker(rank, world, port, data_dir, out_dir):
    dist = _init(rank, world, port)
    corp = _corpora()
    out = _run_api(_base_cfg(data_dir, "rows"), corp)
    from legal_rag_amd.retrieval.vector_store import ShardedFlatIPIndex, VectorStore
    stores = [s for s in VectorStore._instances_by_key.values()]
    assert stores and all(isinstance(s.index, ShardedFlatIPIndex) for s in stores)
    # only this rank's block is resident: local row count == its share of the corpus
    out["resident_rows"] = sorted(int(s.index.native.ntotal) for s in stores)
    Path(out_dir, f"api_{rank}.json").write_text(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def test_api_sharded_rows_equal_unsharded_on_full_fixtures_with_cross_shard_ties(tmp_path):
    from legal_rag_amd import encoders, text
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index
    from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index
    from oracle import bm25 as OB
    from oracle import dense as OD
    from oracle import maxsim as OM
    data = tmp_path / "data"
    corp = _corpora()
    base = _base_cfg(data, None)
    for lang, chunks in corp.items():
        cfg = base.with_lang(lang)
        build_faiss_index(cfg, chunks)
        build_bm25_index(cfg, chunks)
        build_colbert_index(cfg, chunks)
    mp.spawn(_api_worker, args=(2, _free_port(), str(data), str(tmp_path)), nprocs=2, join=True)
    r0 = json.loads((tmp_path / "api_0.json").read_text())
    r1 = json.loads((tmp_path / "api_1.json").read_text())
    n_en, n_zh = len(corp["en"]), len(corp["zh"])
    assert r0.pop("resident_rows") == sorted([n_en // 2, n_zh // 2])
    assert r1.pop("resident_rows") == sorted([n_en - n_en // 2, n_zh - n_zh // 2])
    assert r0 == r1  # identical on every rank, to the last bit of every score and breakdown value
    un = _run_api(base, corp)  # the unsharded product in this process
    for lang in ("zh", "en"):
        for kind in ("single", "batch"):
            for got, exp in zip(r0[lang][kind], un[lang][kind]):
                assert [h["id"] for h in got] == [h["id"] for h in exp], (lang, kind)
                assert [h["source"] for h in got] == [h["source"] for h in exp]
                assert np.allclose([h["score"] for h in got], [h["score"] for h in exp], rtol=0, atol=2e-6)
        assert any(h["source"] == "rerank" for hits in r0[lang]["batch"] for h in hits)
        # BM25 is bit-exact across the split; dense / MaxSim are the same per-row arithmetic
        assert r0[lang]["channels"]["bm25"] == [list(x) for x in un[lang]["channels"]["bm25"]]
        for ch in ("dense", "colbert"):
            assert [i for i, _ in r0[lang]["channels"][ch]] == [i for i, _ in un[lang]["channels"][ch]]
    # ---- per channel against the ORACLE of the unsharded corpus, on the query whose two best documents are the
    # identical pair on either side of the shard boundary: lower global id first, in every channel
    chunks = corp["en"]
    q = _queries(corp)["en"][-1]
    lo_id, hi_id = chunks[TIE_LO].id, chunks[len(chunks) // 2 + 3].id
    emb, te = encoders.HashingEmbedder(768), encoders.HashingTokenEmbedder()
    ds, di = OD.flatip_topk(emb.encode([c.text for c in chunks]), emb.encode_queries([q]), 10)
    ob = OB.BM25Okapi([OB.tokenize_en(c.text) for c in chunks])
    b = OB.search(ob, text.jieba_cut(q), 10)
    mats = [te.encode_doc(c.text.strip()) for c in chunks]
    ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])])
    cs, ci = OM.maxsim_topk(te.encode_query(q.strip())[None], np.concatenate(mats), ptr, 10)
    got = r0["en"]["channels"]
    assert [i for i, _ in got["dense"]] == [chunks[i].id for i in di[0]]
    assert np.allclose([s for _, s in got["dense"]], ds[0], atol=1e-4)
    assert got["bm25"] == [[chunks[i].id, s] for i, s in b]
    assert [i for i, _ in got["colbert"]] == [chunks[i].id for i in ci[0]]
    assert np.allclose([s for _, s in got["colbert"]], cs[0], atol=1e-4)
    for ch in ("dense", "colbert"):
        ids = [i for i, _ in got[ch]]
        assert ids[:2] == [lo_id, hi_id], (ch, ids[:3])
        assert got[ch][0][1] == got[ch][1][1]  # exactly tied scores across the shard boundary
    bm_ids = [i for i, _ in got["bm25"]]
    assert bm_ids.index(lo_id) + 1 == bm_ids.index(hi_id) and got["bm25"][bm_ids.index(lo_id)][1] == got["bm25"][bm_ids.index(hi_id)][1]
