#!/usr/bin/env python3
"""Index build CLI — counterpart of the reference's scripts/build_index.py
(same flags :19-63, same per-language loop :79-112, same version activation
:114-119) over this build's builders.

    python scripts/build_index.py --data-dir data [--only-bm25] [--index-version v1 --activate]
"""
from __future__ import annotations

import argparse
import logging
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from legal_rag_amd.config import AppConfig  # noqa: E402
from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index  # noqa: E402
from legal_rag_amd.retrieval.builders.colbert_builder import build_colbert_index  # noqa: E402
from legal_rag_amd.retrieval.builders.faiss_builder import build_faiss_index  # noqa: E402
from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir  # noqa: E402

logger = logging.getLogger("build_index")


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Build dense + BM25 (+ ColBERT) indexes for the MI355X retrieval engine.")
    g = p.add_mutually_exclusive_group()
    p.add_argument("--data-dir", default="data")
    p.add_argument("--encoder-backend", default="auto", choices=["auto", "transformers", "hashing"])
    p.add_argument("--hnsw-m", type=int, default=32, help="accepted for compatibility; the scan is exact")
    p.add_argument("--hnsw-ef-construction", type=int, default=200, help="accepted for compatibility")
    p.add_argument("--hnsw-ef-search", type=int, default=128, help="accepted for compatibility")
    g.add_argument("--only-colbert", action="store_true")
    g.add_argument("--only-faiss", action="store_true")
    g.add_argument("--only-bm25", action="store_true")
    p.add_argument("--no-faiss", action="store_true")
    p.add_argument("--no-bm25", action="store_true")
    p.add_argument("--no-colbert", action="store_true")
    p.add_argument("--index-version", type=str, default="")
    p.add_argument("--activate", action="store_true")
    return p.parse_args(argv)


def main(argv=None) -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    a = parse_args(argv)
    version = (a.index_version or "").strip() or None
    base = AppConfig.for_data_dir(a.data_dir, "zh", index_version=version)
    base.retrieval.encoder_backend = a.encoder_backend
    chunks = load_chunks_from_dir(base.retrieval.processed_dir, base.retrieval.processed_glob)
    logger.info("Loaded %d law chunks from %s/%s", len(chunks), base.retrieval.processed_dir,
                base.retrieval.processed_glob)
    by_lang = {}
    for c in chunks:
        by_lang.setdefault((getattr(c, "lang", None) or "zh").strip().lower(), []).append(c)
    if not by_lang:
        logger.error("No chunks found to index.")
        return
    for lang, lang_chunks in sorted(by_lang.items()):
        cfg = base.with_lang(lang, index_version=version)
        logger.info("Building indexes for lang=%s (chunks=%d)", lang, len(lang_chunks))
        if a.only_colbert:
            build_colbert_index(cfg, lang_chunks)
            continue
        if a.only_faiss:
            build_faiss_index(cfg, lang_chunks)
            continue
        if a.only_bm25:
            build_bm25_index(cfg, lang_chunks)
            continue
        if not a.no_faiss:
            build_faiss_index(cfg, lang_chunks)
        if not a.no_bm25:
            build_bm25_index(cfg, lang_chunks)
        if not a.no_colbert:
            try:
                build_colbert_index(cfg, lang_chunks)
            except Exception as e:  # noqa: BLE001 - tolerated like the reference (:108-112)
                print(f"Warning: ColBERT index build failed for lang={lang}, continuing without it.\nReason: {e}")
    if version and a.activate:
        for lang in sorted(by_lang):
            root = Path(base.with_lang(lang).paths.index_dir)
            if not (root / "versions" / version).exists():
                raise FileNotFoundError(f"index version not found: {version}")
            (root / "ACTIVE").write_text(version, encoding="utf-8")


if __name__ == "__main__":
    main()
