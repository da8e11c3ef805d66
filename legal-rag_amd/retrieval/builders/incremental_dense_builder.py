"""Incremental dense add (legalrag/retrieval/builders/incremental_dense_builder.py:18-78).

`add_jsonl(path)`: load the incoming chunks, take the index file lock
(`<faiss.index>.lock` with the suffix replaced, as the reference), reload the
store, drop ids that already exist, embed the rest as passages, append the rows
to the resident matrix (`amdr_dense_add` — serialised against concurrent searches
inside the library), append the meta lines FIRST, extend the in-memory chunk
list, then persist the index file.  Returns the number of chunks added."""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import List

from filelock import FileLock

from ... import artifacts
from ...schemas import LawChunk
from ..vector_store import VectorStore

logger = logging.getLogger(__name__)


class IncrementalDenseBuilder:
    def __init__(self, cfg):
        self.cfg = cfg
        self.vs = VectorStore.from_config(cfg)

    @staticmethod
    def _load_jsonl_chunks(jsonl_path: Path) -> List[LawChunk]:
        fields = set(LawChunk.model_fields)
        out: List[LawChunk] = []
        with jsonl_path.open("r", encoding="utf-8") as f:
            for line in f:
                if line.strip():
                    out.append(LawChunk(**{k: v for k, v in json.loads(line).items() if k in fields}))
        return out

    def add_jsonl(self, jsonl_path) -> int:
        jsonl_path = Path(jsonl_path)
        if not jsonl_path.exists():
            logger.error("[index] jsonl not found: %s", jsonl_path)
            raise FileNotFoundError(jsonl_path)
        incoming = self._load_jsonl_chunks(jsonl_path)
        if not incoming:
            logger.warning("[index] empty jsonl, skip: %s", jsonl_path)
            return 0
        lock_path = Path(self.cfg.retrieval.faiss_index_file).with_suffix(".lock")
        with FileLock(str(lock_path)):
            self.vs.load()
            exist_ids = {c.id for c in self.vs.chunks}
            new_chunks: List[LawChunk] = []
            for c in incoming:  # also de-duplicates inside the incoming file
                if c.id not in exist_ids:
                    exist_ids.add(c.id)
                    new_chunks.append(c)
            added = len(new_chunks)
            logger.info("[index] dedup done: incoming=%d added=%d", len(incoming), added)
            if added == 0:
                return 0
            vecs = self.vs._embed([c.text for c in new_chunks]).astype("float32")
            self.vs.index.add(vecs)
            self.vs.meta_path.parent.mkdir(parents=True, exist_ok=True)
            with self.vs.meta_path.open("a", encoding="utf-8") as f:
                for c in new_chunks:
                    f.write(c.model_dump_json() + "\n")
            self.vs.chunks.extend(new_chunks)
            n = self.vs.index.ntotal
            artifacts.write_faiss_flat_ip(self.vs.index_path, self.vs.index.reconstruct_n(0, n))
            # the files this process just wrote ARE the resident state: skip the mtime reload
            self.vs._index_mtime = self.vs.index_path.stat().st_mtime
            self.vs._meta_mtime = self.vs.meta_path.stat().st_mtime
        logger.info("[index] incremental add done: added=%d jsonl=%s", added, jsonl_path)
        return added
