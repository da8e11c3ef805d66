"""Corpus loading with id de-duplication (legalrag/retrieval/corpus_loader.py:22-39)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Iterable, List, Set

from ..schemas import LawChunk

_FIELDS = set(LawChunk.model_fields)


def _chunk(obj: dict) -> LawChunk:
    return LawChunk(**{k: v for k, v in obj.items() if k in _FIELDS})


def iter_chunks_from_dir(processed_dir: str, pattern: str = "*.jsonl") -> Iterable[LawChunk]:
    for fp in sorted(Path(processed_dir).glob(pattern)):
        if fp.is_dir():
            continue
        with fp.open("r", encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if line:
                    yield _chunk(json.loads(line))


def load_chunks_from_dir(processed_dir: str, pattern: str = "*.jsonl") -> List[LawChunk]:
    seen: Set[str] = set()
    out: List[LawChunk] = []
    for fp in sorted(Path(processed_dir).glob(pattern)):
        if not fp.is_file():
            continue
        with fp.open("r", encoding="utf-8") as f:
            for line in f:
                if not line.strip():
                    continue
                c = _chunk(json.loads(line))
                if c.id in seen:
                    continue
                seen.add(c.id)
                out.append(c)
    return out
