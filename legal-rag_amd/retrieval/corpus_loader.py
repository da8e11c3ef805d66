"""Corpus loading (counterpart of legalrag/retrieval/corpus_loader.py:10-39).

`iter_chunks_from_dir` streams every chunk of every `*.jsonl` file in name
order; `load_chunks_from_dir` keeps the FIRST chunk seen for each id (UCC-en:
592 parsed records -> 591 chunks, the duplicate is `ucc_4A.txt::4A-102`).
Extra keys that the preprocessing step emits (part, subpart, article_key) are
ignored, as pydantic does for the reference's LawChunk."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Iterator, List

from ..schemas import LawChunk

_KEEP = frozenset(LawChunk.model_fields)


def chunk_from_record(rec: dict) -> LawChunk:
    return LawChunk(**{k: rec[k] for k in rec.keys() & _KEEP})


def read_jsonl_chunks(path: Path) -> Iterator[LawChunk]:
    with Path(path).open(encoding="utf-8") as fh:
        for raw in fh:
            if raw.strip():
                yield chunk_from_record(json.loads(raw))


def _files(processed_dir: str, pattern: str) -> List[Path]:
    return [p for p in sorted(Path(processed_dir).glob(pattern)) if p.is_file()]


def iter_chunks_from_dir(processed_dir: str, pattern: str = "*.jsonl") -> Iterator[LawChunk]:
    for path in _files(processed_dir, pattern):
        yield from read_jsonl_chunks(path)


def load_chunks_from_dir(processed_dir: str, pattern: str = "*.jsonl") -> List[LawChunk]:
    first_by_id: Dict[str, LawChunk] = {}
    for chunk in iter_chunks_from_dir(processed_dir, pattern):
        first_by_id.setdefault(chunk.id, chunk)  # dicts keep insertion order
    return list(first_by_id.values())
