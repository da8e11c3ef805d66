"""Incremental BM25 rebuild (legalrag/retrieval/builders/incremental_bm25_builder.py:19-82).

BM25 statistics are corpus-global, so "incremental" means: existing chunks (from
the current bm25.pkl, [] if unreadable) + new ids -> full re-fit -> atomic
replace of the pickle.  NOTE the reference tokenises EVERY chunk with jieba.cut
here, even English ones (unlike build_bm25_index, which lower-cases English with
a regex): that quirk is kept so a re-built index equals the reference's."""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import List

from ... import artifacts, text
from ...bm25_model import BM25Okapi
from ...schemas import LawChunk

logger = logging.getLogger(__name__)


class IncrementalBM25Builder:
    def __init__(self, cfg):
        self.cfg = cfg

    @staticmethod
    def _load_jsonl_chunks(jsonl_path: Path) -> List[LawChunk]:
        fields = set(LawChunk.model_fields)
        out: List[LawChunk] = []
        with jsonl_path.open("r", encoding="utf-8") as f:
            for line in f:
                if line.strip():
                    out.append(LawChunk(**{k: v for k, v in json.loads(line).items() if k in fields}))
        return out

    def _load_existing(self) -> List[LawChunk]:
        p = Path(self.cfg.retrieval.bm25_index_file)
        if not p.exists():
            return []
        try:
            return artifacts.read_bm25_pickle(p)[1]
        except Exception:  # noqa: BLE001 - unreadable index == start over, as the reference
            return []

    def add_jsonl(self, jsonl_path) -> int:
        jsonl_path = Path(jsonl_path)
        if not jsonl_path.exists():
            logger.error("[bm25] jsonl not found: %s", jsonl_path)
            raise FileNotFoundError(jsonl_path)
        incoming = self._load_jsonl_chunks(jsonl_path)
        if not incoming:
            logger.warning("[bm25] empty jsonl, skip: %s", jsonl_path)
            return 0
        existing = self._load_existing()
        exist_ids = {c.id for c in existing}
        new_chunks = [c for c in incoming if c.id not in exist_ids]
        if not new_chunks:
            logger.info("[bm25] no new chunks to add: incoming=%d", len(incoming))
            return 0
        all_chunks = existing + new_chunks
        bm25 = BM25Okapi([text.jieba_cut(c.text) for c in all_chunks])
        artifacts.write_bm25_pickle(Path(self.cfg.retrieval.bm25_index_file), bm25, all_chunks)
        logger.info("[bm25] incremental add done: added=%d total=%d", len(new_chunks), len(all_chunks))
        return len(new_chunks)
