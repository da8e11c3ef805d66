"""BM25 index builder (legalrag/retrieval/builders/bm25_builder.py:22-53):
tokenise (English: lower-cased regex words; otherwise jieba), fit Okapi
statistics, persist {"bm25", "chunks"[dict]} to bm25.pkl.  The language is taken
from the FIRST chunk, as the reference does (:39)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List

from ... import artifacts, text
from ...bm25_model import BM25Okapi
from ...schemas import LawChunk

logger = logging.getLogger(__name__)

_tokenize_en = text.tokenize_en


def tokenize_corpus(chunks: List[LawChunk]) -> List[List[str]]:
    lang = (getattr(chunks[0], "lang", None) or "zh").strip().lower() if chunks else "zh"
    if lang == "en":
        return [_tokenize_en(c.text) for c in chunks]
    return [text.jieba_cut(c.text) for c in chunks]


def build_bm25_index(cfg, chunks: List[LawChunk]) -> None:
    bm25_path = Path(cfg.retrieval.bm25_index_file)
    logger.info("[BM25] building (docs=%d) -> %s", len(chunks), bm25_path)
    bm25 = BM25Okapi(tokenize_corpus(chunks))
    artifacts.write_bm25_pickle(bm25_path, bm25, chunks)
    logger.info("[BM25] saved -> %s", bm25_path)
