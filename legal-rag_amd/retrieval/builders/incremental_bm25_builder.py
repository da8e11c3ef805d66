"""Incremental BM25 rebuild (counterpart of
legalrag/retrieval/builders/incremental_bm25_builder.py:19-82).

BM25 statistics are corpus-global, so "incremental" is: chunks of the current
bm25.pkl (none if it is missing or unreadable) + the new ids -> full re-fit ->
atomic replace.  The reference tokenises EVERY chunk with jieba.cut on this path,
English included (build_bm25_index lower-cases English with a regex instead);
the quirk is kept so that a re-built index equals the reference's."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List

from ... import artifacts, text
from ...bm25_model import BM25Okapi
from ...schemas import LawChunk
from ._incremental import incoming_chunks

logger = logging.getLogger(__name__)


class IncrementalBM25Builder:
    def __init__(self, cfg):
        self.cfg = cfg

    def _current_chunks(self) -> List[LawChunk]:
        path = Path(self.cfg.retrieval.bm25_index_file)
        try:
            return artifacts.read_bm25_pickle(path)[1] if path.exists() else []
        except Exception:  # noqa: BLE001 - an unreadable index means start over, as the reference does
            return []

    def add_jsonl(self, jsonl_path) -> int:
        batch = incoming_chunks(jsonl_path, logger, "bm25")
        if not batch:
            return 0
        kept = self._current_chunks()
        fresh = [c for c in batch if c.id not in {k.id for k in kept}]
        if not fresh:
            logger.info("[bm25] no new chunks to add: incoming=%d", len(batch))
            return 0
        corpus = kept + fresh
        mode = text.cfg_mode(self.cfg)
        artifacts.write_bm25_pickle(Path(self.cfg.retrieval.bm25_index_file),
                                    BM25Okapi([text.jieba_cut(c.text, mode) for c in corpus]), corpus,
                                    tokenizer=text.tokenizer_id(mode))
        logger.info("[bm25] incremental add done: added=%d total=%d", len(fresh), len(corpus))
        return len(fresh)
