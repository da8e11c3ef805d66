"""Configuration for the retrieval path.

The retrievers read `cfg.retrieval.*` by getattr with the reference's defaults
(hybrid_retriever.py:289,309,324-336,398-405), so the reference's own pydantic
AppConfig — or any duck-typed object, as the reference's tests use
(tests/test_retrieval.py:17-33) — can be passed unchanged.  This module supplies
a standalone equivalent for the knobs the path reads (legalrag/config.py:54-129)
and the per-language index layout (`with_lang`, config.py:159-204,248-274) with
the IndexRegistry convention `<index_root>/ACTIVE` + `versions/<v>/`
(legalrag/index/registry.py:19-53).  Unlike the reference, constructing a config
never creates directories.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field, replace
from pathlib import Path
from typing import Dict, List, Optional


@dataclass
class RetrievalConfig:
    processed_file: str = "processed/law_zh.jsonl"
    processed_dir: str = "data/processed"
    processed_glob: str = "*.jsonl"
    # dense
    faiss_index_file: str = "index/faiss/faiss.index"
    faiss_meta_file: str = "index/faiss/faiss_meta.jsonl"
    embedding_model: str = "BAAI/bge-base-zh-v1.5"
    embedding_model_zh: str = "BAAI/bge-base-zh-v1.5"
    embedding_model_en: str = "BAAI/bge-base-en-v1.5"
    hnsw_m: int = 64                 # accepted for CLI compatibility; the scan is exact
    hnsw_ef_construction: int = 400
    hnsw_ef_search: int = 512
    # sparse
    bm25_index_file: str = "index/bm25.pkl"
    # graph (config.py:74-88; the channel runs only for GRAPH_AUGMENTED routing decisions and
    # switches itself off when the graph file is missing)
    enable_graph: bool = True
    graph_seed_k: int = 30
    graph_walk_depths: Dict[str, int] = field(default_factory=lambda: {
        "defined_by": 4, "defines_term": 3, "cite": 1, "cited_by": 1, "prev": 2, "next": 2, "default": 2})
    graph_limit: int = 800
    graph_weight: float = 0.2
    graph_rel_types: Optional[List[str]] = None
    # control
    top_k: int = 10
    bm25_weight: float = 0.4
    dense_weight: float = 0.6
    min_final_score: float = 0.2
    # colbert
    enable_colbert: bool = True
    colbert_index_path: str = ""
    colbert_meta_file: str = "index/colbert/colbert_meta.jsonl"
    colbert_weight: float = 0.35
    colbert_index_name: str = "law"
    colbert_index_name_zh: str = "law_zh"
    colbert_index_name_en: str = "law_en"
    colbert_model_name: str = "jinaai/jina-colbert-v2"
    colbert_experiment: str = "experiment"
    colbert_nranks: int = 1
    colbert_nbits: int = 4
    colbert_doc_maxlen: int = 220
    colbert_kmeans_niters: int = 10
    # rerank
    enable_rerank: bool = True
    rerank_top_n: int = 30
    rrf_alpha: float = 0.5
    rerank_beta: float = 0.35
    rerank_ce_model: str = "BAAI/bge-reranker-v2-m3"
    rerank_use_llm: bool = False
    # fusion
    fusion_method: str = "rrf_norm_blend"
    rrf_k: int = 60
    # --- build-specific knobs (no reference counterpart) ---
    device: int = 0                      # HIP device ordinal of this process
    encoder_backend: str = "auto"        # "auto" | "transformers" | "hashing" (deterministic stand-in)
    rerank_fp16: bool = False            # cross-encoder in half precision (the reference's CrossEncoder runs fp32)
    shard: Optional[str] = None          # "rows": this process holds the row block of its torch.distributed rank in
                                         # HBM and every search all-gathers the per-shard top-k (retrieval/sharding.py)
    zh_tokenizer: str = "jieba"          # "jieba" (raises if Han text meets no segmenter) | "char" (explicit
                                         # opt-in to the inexact one-character stand-in, text.py)


@dataclass
class PathsConfig:
    data_dir: str = "data"
    processed_dir: str = "data/processed"
    index_dir: str = "data/index"
    law_jsonl: str = "data/processed/law_zh.jsonl"
    graph_dir: str = "data/graph"
    law_graph_jsonl: str = "data/graph/law_graph_zh.jsonl"


def active_index_dir(index_root: Path, version: Optional[str] = None) -> Path:
    """IndexRegistry.active_index_dir / ensure_version_dir without the mkdir."""
    if version:
        return index_root / "versions" / version
    active = index_root / "ACTIVE"
    if active.exists():
        v = active.read_text(encoding="utf-8").strip()
        if v and (index_root / "versions" / v).exists():
            return index_root / "versions" / v
    return index_root


@dataclass
class AppConfig:
    paths: PathsConfig = field(default_factory=PathsConfig)
    retrieval: RetrievalConfig = field(default_factory=RetrievalConfig)

    @classmethod
    def for_data_dir(cls, data_dir: str, lang: str = "zh", index_version: Optional[str] = None) -> "AppConfig":
        cfg = cls(paths=PathsConfig(data_dir=str(data_dir)))
        return cfg.with_lang(lang, index_version=index_version)

    def with_lang(self, lang: str, index_version: Optional[str] = None) -> "AppConfig":
        lang_key = (lang or "zh").strip().lower()
        data_dir = Path(self.paths.data_dir)
        processed = data_dir / "processed"
        index_root = data_dir / "index" / lang_key
        version = index_version or os.getenv("LEGALRAG_INDEX_VERSION", "").strip() or None
        act = active_index_dir(index_root, version)
        r = replace(self.retrieval)
        r.processed_dir = str(processed)
        r.processed_file = str(processed / f"law_{lang_key}.jsonl")
        r.faiss_index_file = str(act / "faiss" / "faiss.index")
        r.faiss_meta_file = str(act / "faiss" / "faiss_meta.jsonl")
        r.bm25_index_file = str(act / "bm25.pkl")
        r.colbert_index_path = str(act / "colbert")
        r.colbert_meta_file = str(act / "colbert" / "colbert_meta.jsonl")
        if lang_key == "en":
            r.embedding_model = r.embedding_model_en
            r.colbert_index_name = r.colbert_index_name_en
        else:
            r.embedding_model = r.embedding_model_zh
            r.colbert_index_name = r.colbert_index_name_zh
        p = PathsConfig(data_dir=str(data_dir), processed_dir=str(processed), index_dir=str(index_root),
                        law_jsonl=str(processed / f"law_{lang_key}.jsonl"), graph_dir=str(data_dir / "graph"),
                        law_graph_jsonl=str(data_dir / "graph" / f"law_graph_{lang_key}.jsonl"))
        return AppConfig(paths=p, retrieval=r)
