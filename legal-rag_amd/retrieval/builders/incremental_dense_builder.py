"""Incremental dense add (counterpart of
legalrag/retrieval/builders/incremental_dense_builder.py:18-78).

Under the index file lock (`faiss.index` with the suffix replaced by `.lock`,
the reference's name): reload the store, keep only ids it does not hold yet,
embed them as passages, append the rows to the matrix resident in HBM
(`amdr_dense_add` is serialised against concurrent searches inside the library),
append the meta lines BEFORE rewriting the index file (a reader must never see
an index row without its chunk), then persist the index."""
from __future__ import annotations

import logging
from pathlib import Path

from filelock import FileLock

from ... import artifacts
from ..vector_store import VectorStore
from ._incremental import incoming_chunks, unseen

logger = logging.getLogger(__name__)


class IncrementalDenseBuilder:
    def __init__(self, cfg):
        self.cfg = cfg
        self.vs = VectorStore.from_config(cfg)

    def add_jsonl(self, jsonl_path) -> int:
        batch = incoming_chunks(jsonl_path, logger, "index")
        if not batch:
            return 0
        vs = self.vs
        with FileLock(str(Path(self.cfg.retrieval.faiss_index_file).with_suffix(".lock"))):
            vs.load()
            fresh = unseen(batch, {c.id for c in vs.chunks})
            logger.info("[index] dedup done: incoming=%d added=%d", len(batch), len(fresh))
            if not fresh:
                return 0
            vs.index.add(vs._embed([c.text for c in fresh]).astype("float32"))
            vs.meta_path.parent.mkdir(parents=True, exist_ok=True)
            with vs.meta_path.open("a", encoding="utf-8") as meta:
                meta.writelines(c.model_dump_json() + "\n" for c in fresh)
            vs.chunks.extend(fresh)
            artifacts.write_faiss_flat_ip(vs.index_path, vs.index.reconstruct_n(0, vs.index.ntotal))
            # what this process just wrote IS the resident state: no mtime-triggered reload
            vs._index_mtime = vs.index_path.stat().st_mtime
            vs._meta_mtime = vs.meta_path.stat().st_mtime
        logger.info("[index] incremental add done: added=%d jsonl=%s", len(fresh), jsonl_path)
        return len(fresh)
