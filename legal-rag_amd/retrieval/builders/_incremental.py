"""Shared pieces of the incremental (ingest-time) builders."""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, List, Set

from ...schemas import LawChunk
from ..corpus_loader import read_jsonl_chunks


def incoming_chunks(jsonl_path, log, tag: str) -> List[LawChunk]:
    """Chunks of an ingest JSONL; FileNotFoundError if the file is missing."""
    path = Path(jsonl_path)
    if not path.exists():
        log.error("[%s] jsonl not found: %s", tag, path)
        raise FileNotFoundError(path)
    chunks = list(read_jsonl_chunks(path))
    if not chunks:
        log.warning("[%s] empty jsonl, skip: %s", tag, path)
    return chunks


def unseen(chunks: Iterable[LawChunk], known_ids: Set[str]) -> List[LawChunk]:
    """Chunks whose id is not in `known_ids` (nor earlier in `chunks`); order kept."""
    seen = set(known_ids)
    fresh: List[LawChunk] = []
    for c in chunks:
        if c.id not in seen:
            seen.add(c.id)
            fresh.append(c)
    return fresh
