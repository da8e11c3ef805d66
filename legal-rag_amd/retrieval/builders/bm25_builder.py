"""BM25 index builder (legalrag/retrieval/builders/bm25_builder.py:22-53):
tokenise (English: lower-cased regex words; otherwise jieba), fit Okapi
statistics, persist {"bm25", "chunks"[dict]} to bm25.pkl.  The language is taken
from the FIRST chunk, as the reference does (:39).

Han text without jieba is never tokenised silently (text.py): the build raises
`text.ZhTokenizerUnavailable` unless the caller passes pre-tokenised documents
(`tokens=`), registers a segmenter, or opts in with cfg.retrieval.zh_tokenizer =
"char".  The id of the segmenter used is stored in bm25.pkl ("tokenizer")."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

from ... import artifacts, text
from ...bm25_model import BM25Okapi
from ...schemas import LawChunk

logger = logging.getLogger(__name__)

_tokenize_en = text.tokenize_en


def corpus_lang(chunks: Sequence[LawChunk]) -> str:
    return (getattr(chunks[0], "lang", None) or "zh").strip().lower() if chunks else "zh"


def tokenize_corpus(chunks: List[LawChunk], mode: Optional[str] = None) -> Tuple[List[List[str]], str]:
    """-> (token lists, tokenizer id)."""
    if corpus_lang(chunks) == "en":
        return [_tokenize_en(c.text) for c in chunks], "en_regex"
    return [text.jieba_cut(c.text, mode) for c in chunks], text.tokenizer_id(mode)


def build_bm25_index(cfg, chunks: List[LawChunk], tokens: Optional[Sequence[Sequence[str]]] = None,
                     tokenizer: str = "pretokenized") -> None:
    """`tokens`: one token list per chunk, produced by the caller (e.g. with jieba on another
    machine) — the exact path when jieba is not importable here; `tokenizer` names it."""
    bm25_path = Path(cfg.retrieval.bm25_index_file)
    logger.info("[BM25] building (docs=%d) -> %s", len(chunks), bm25_path)
    if tokens is not None:
        if len(tokens) != len(chunks):
            raise ValueError(f"build_bm25_index: {len(tokens)} token lists for {len(chunks)} chunks")
        docs, tok_id = [list(t) for t in tokens], str(tokenizer)
    else:
        docs, tok_id = tokenize_corpus(chunks, text.cfg_mode(cfg))
    bm25 = BM25Okapi(docs)
    artifacts.write_bm25_pickle(bm25_path, bm25, chunks, tokenizer=tok_id)
    logger.info("[BM25] saved -> %s (tokenizer=%s)", bm25_path, tok_id)
