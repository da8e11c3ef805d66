#!/usr/bin/env python3
"""cProfile of HybridRetriever.search_batch / search_batch_arrays on the UCC-en fixture (GPU box): where the
host time of the Python API goes.    python scripts/profile_search_batch.py"""
import cProfile
import pstats
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from legal_rag_amd.config import AppConfig  # noqa: E402
from legal_rag_amd.evaluation import synthetic_queries  # noqa: E402
from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index  # noqa: E402
from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index  # noqa: E402
from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir  # noqa: E402
from legal_rag_amd.retrieval.hybrid_retriever import HybridRetriever  # noqa: E402

with tempfile.TemporaryDirectory(prefix="amdr_prof_") as tmp:
    cfg = AppConfig.for_data_dir(tmp, "en")
    cfg.retrieval.encoder_backend = "hashing"
    cfg.retrieval.device = 0
    cfg.retrieval.enable_graph = False
    cfg.retrieval.enable_colbert = cfg.retrieval.enable_rerank = False
    chunks = load_chunks_from_dir(str(ROOT / "tests" / "golden" / "corpus"), "law_en.jsonl")
    build_faiss_index(cfg, chunks)
    build_bm25_index(cfg, chunks)
    qs = [q for q, _, _ in synthetic_queries(chunks, seed=0)]
    r = HybridRetriever(cfg)
    r.search_batch(qs[:64], top_k=10)
    for name, fn in (("search_batch", lambda: r.search_batch(qs, top_k=10)),
                     ("search_batch_arrays", lambda: r.search_batch_arrays(qs * 4, top_k=10))):
        pr = cProfile.Profile()
        pr.enable()
        fn()
        pr.disable()
        print("=====", name)
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
