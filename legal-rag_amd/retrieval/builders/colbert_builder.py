"""ColBERT index builder (legalrag/retrieval/builders/colbert_builder.py:55-136):
writes colbert_meta.jsonl ({"pid", "chunk"}, pid == row) and, in place of the
PLAID index colbert-ai would produce, this build's token store (fp32 token
embeddings + doc_ptr) inside the same <root>/<experiment>/indexes/<name>/
directory.  Serialised by a FileLock on the index directory like the reference."""
from __future__ import annotations

from pathlib import Path
from typing import List

import numpy as np
from filelock import FileLock

from ... import artifacts
from ...schemas import LawChunk
from ..colbert_retriever import get_token_encoder


def build_colbert_index(cfg, chunks: List[LawChunk], override: bool = False) -> Path:
    rcfg = cfg.retrieval
    if not bool(getattr(rcfg, "enable_colbert", False)):
        raise RuntimeError("ColBERT is disabled: set cfg.retrieval.enable_colbert=True")
    index_path = Path(str(getattr(rcfg, "colbert_index_path")))
    index_name = str(getattr(rcfg, "colbert_index_name"))
    meta_file = Path(str(getattr(rcfg, "colbert_meta_file")))
    experiment = str(getattr(rcfg, "colbert_experiment"))
    doc_maxlen = int(getattr(rcfg, "colbert_doc_maxlen", 220))
    docs = [(getattr(c, "text", "") or "").strip() for c in chunks]
    if not any(docs):
        raise RuntimeError("All chunks are empty; cannot build ColBERT index.")
    enc = get_token_encoder(getattr(rcfg, "colbert_model_name", None), str(getattr(rcfg, "encoder_backend", "auto")),
                            doc_maxlen, device=f"cuda:{int(getattr(rcfg, 'device', 0))}")
    index_path.mkdir(parents=True, exist_ok=True)
    with FileLock(str(index_path / ".colbert_build.lock")):
        out_dir = artifacts.colbert_index_dir(str(index_path), experiment, index_name)
        if (out_dir / "amdr_tokens.npz").exists() and not override:
            # colbert-ai's overwrite=False default reuses an existing index of the same name
            artifacts.write_colbert_meta(meta_file, chunks)
            return out_dir.resolve()
        artifacts.write_colbert_meta(meta_file, chunks)
        mats = [np.asarray(enc.encode_doc(d), dtype=np.float32) for d in docs]
        doc_ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
        artifacts.write_token_store(out_dir, np.concatenate(mats, axis=0), doc_ptr)
    return out_dir.resolve()
