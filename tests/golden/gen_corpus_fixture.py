#!/usr/bin/env python3
"""Generate the law_{en,zh}.jsonl corpus fixtures (DATA, not source).

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's own text->JSONL parser
(scripts/preprocess_law.py: parse_by_lines / parse_by_scan_fallback, the
selection rule of main() at :507-523) over the public-law raw text under
data/raw/ and writes the parsed records to tests/golden/corpus/.  Nothing from
/root/reference is copied except the parsed law text itself (public statutes).

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/gen_corpus_fixture.py
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

REF = Path("/root/reference")
sys.path.insert(0, str(REF))
sys.dont_write_bytecode = True

from scripts import preprocess_law as pl  # noqa: E402

OUT = Path(__file__).resolve().parent / "corpus"


def main() -> int:
    raw_dir = REF / "data" / "raw"
    txt_files = sorted(raw_dir.rglob("*.txt"))
    all_records = []
    for p in txt_files:
        text = pl._read_text(p)
        lang = pl.detect_lang(text)
        law_name = "Uniform Commercial Code" if lang == "en" else "中华人民共和国民法典"
        recs_line = pl.parse_by_lines(text, source=p.name, law_name=law_name)
        recs_scan = pl.parse_by_scan_fallback(text, source=p.name, law_name=law_name)
        if recs_scan and (len(recs_line) < 10 or len(recs_scan) > len(recs_line)):
            recs = recs_scan
        else:
            recs = recs_line
        print(f"{p.name}: line={len(recs_line)} scan={len(recs_scan)} -> {len(recs)}")
        all_records.extend(recs)
    by_lang = {}
    for r in all_records:
        by_lang.setdefault(str(r.get("lang") or "zh").strip().lower(), []).append(r)
    OUT.mkdir(parents=True, exist_ok=True)
    for lang, recs in by_lang.items():
        out = OUT / f"law_{lang}.jsonl"
        with out.open("w", encoding="utf-8") as f:
            for r in recs:
                f.write(json.dumps(r, ensure_ascii=False) + "\n")
        print(f"wrote {len(recs)} records -> {out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
