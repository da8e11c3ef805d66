#!/usr/bin/env python3
"""cProfile of HybridRetriever.search (one query per call) on the UCC-en fixture (GPU box):
    python scripts/profile_search_single.py            dense + BM25
    python scripts/profile_search_single.py full       the reference's default: + ColBERT + rerank (stand-in CE)"""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from legal_rag_amd.config import AppConfig  # noqa: E402
from legal_rag_amd.evaluation import synthetic_queries  # noqa: E402
from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index  # noqa: E402
from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index  # noqa: E402
from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index  # noqa: E402
from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir  # noqa: E402
from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever  # noqa: E402

with tempfile.TemporaryDirectory(prefix="amdr_prof_") as tmp:
    cfg = AppConfig.for_data_dir(tmp, "en")
    cfg.retrieval.encoder_backend = "hashing"
    cfg.retrieval.device = 0
    cfg.retrieval.enable_graph = False
    full = len(sys.argv) > 1 and sys.argv[1] == "full"
    cfg.retrieval.rerank_ce_model = "hashing"
    cfg.retrieval.enable_colbert = cfg.retrieval.enable_rerank = full
    chunks = load_chunks_from_dir(str(ROOT / "tests" / "golden" / "corpus"), "law_en.jsonl")
    build_faiss_index(cfg, chunks)
    build_bm25_index(cfg, chunks)
    if full:
        build_colbert_index(cfg, chunks)
    qs = [q for q, _, _ in synthetic_queries(chunks, seed=0)]
    r = HybridRetriever(cfg)
    for q in qs[:30]:
        r.search(q, top_k=10)
    t = time.perf_counter()
    for q in qs[30:330]:
        r.search(q, top_k=10)
    print("mean ms per search():", (time.perf_counter() - t) / 300 * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for q in qs[330:630]:  # queries not seen before: what a caller's stream of distinct questions pays
        r.search(q, top_k=10)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
