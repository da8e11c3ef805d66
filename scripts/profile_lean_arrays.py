#!/usr/bin/env python3
"""Where the time of search_batch_arrays(q_emb=, values=False) goes (9 344 UCC-en queries, GPU box): cProfile + stage stamps."""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

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
    qs = [q for q, _, _ in synthetic_queries(chunks, seed=0)] * 8
    r = HybridRetriever(cfg)
    emb = r.dense.store.embed_device(qs[:1168], is_query=True).repeat(8, 1).contiguous()
    for _ in range(3):
        r.search_batch_arrays(qs, top_k=10, q_emb=emb, values=False)
    ts = []
    for _ in range(9):
        t = time.perf_counter()
        r.search_batch_arrays(qs, top_k=10, q_emb=emb, values=False)
        ts.append((time.perf_counter() - t) * 1e3)
    print("ms per call:", sorted(round(x, 3) for x in ts))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        r.search_batch_arrays(qs, top_k=10, q_emb=emb, values=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
