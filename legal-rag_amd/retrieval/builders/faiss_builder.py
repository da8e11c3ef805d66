"""Dense index builder (legalrag/retrieval/builders/faiss_builder.py:66-104):
embed every chunk's text as a passage (no instruction), write
`faiss/faiss.index` + `faiss/faiss_meta.jsonl`.  The index file is a FAISS
IndexFlatIP container (see artifacts.py): the exact function the reference's
IndexHNSWFlat approximates, loadable by the reference unchanged."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List

import numpy as np

from ... import artifacts, encoders
from ...schemas import LawChunk

logger = logging.getLogger(__name__)


def build_faiss_index(cfg, chunks: List[LawChunk]) -> None:
    rcfg = cfg.retrieval
    model = encoders.get_embedder(str(rcfg.embedding_model), backend=str(getattr(rcfg, "encoder_backend", "auto")),
                                  dim=int(getattr(rcfg, "embedding_dim", 768)))
    texts = [c.text for c in chunks]
    if texts:
        emb = np.asarray(model.encode(texts, batch_size=64, max_length=512)).astype("float32")
    else:
        emb = np.zeros((0, int(model.hidden_size)), dtype="float32")
    artifacts.write_faiss_flat_ip(Path(rcfg.faiss_index_file), emb)
    logger.info("[FAISS] index written: %s (flat inner-product, %d x %d)", rcfg.faiss_index_file, *emb.shape)
    artifacts.write_faiss_meta(Path(rcfg.faiss_meta_file), chunks)
    logger.info("[FAISS] meta written: %s", rcfg.faiss_meta_file)
