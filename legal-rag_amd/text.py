"""Tokenisers of the BM25 channel (host side).

Index side, English: lower-cased regex words (legalrag/retrieval/builders/
bm25_builder.py:18-19,39-41).  Index side, Chinese, and EVERY query regardless
of language: `jieba.cut` (bm25_builder.py:43, bm25_retriever.py:73) — note the
reference does NOT lower-case queries, so capitalised English query words never
match the lower-cased English index; that behaviour is kept.

jieba (`jieba>=0.42.1`, requirements.txt) is absent from the build container.
If it is importable it is used.  Otherwise `jieba_cut` restates jieba 0.42.1's
default mode (cut_all=False, HMM=True) for text WITHOUT Han characters, which
needs no dictionary beyond a handful of ASCII entries (restated from the
published algorithm: jieba/__init__.py `cut`/`__cut_DAG`, finalseg `cut`):
  1. split on runs of [\\u4E00-\\u9FD5a-zA-Z0-9+#&._%-]; between such blocks,
     whitespace (\\r\\n or one \\s char) is emitted as a token and every other
     character as its own token;
  2. inside a block made of single-character DAG steps the whole buffer goes
     through finalseg.cut: runs matching [a-zA-Z0-9]+(?:\\.\\d+)?%? are one token
     each, and each maximal run of the remaining characters is one token.
Han runs cannot be segmented without dict.txt: they are emitted one character
per token and `ZH_EXACT` is False — zh BM25 parity is then UNPINNED.
"""
from __future__ import annotations

import re
from typing import List

try:  # pragma: no cover - not installed in the build container
    import jieba as _jieba
    HAVE_JIEBA = True
except Exception:  # noqa: BLE001
    _jieba = None
    HAVE_JIEBA = False

ZH_EXACT = HAVE_JIEBA

_EN_INDEX_RE = re.compile(r"[A-Za-z0-9]+(?:'[A-Za-z0-9]+)?")
_RE_BLOCK = re.compile(r"([一-鿕a-zA-Z0-9+#&\._%\-]+)", re.U)
_RE_SKIP = re.compile(r"(\r\n|\s)", re.U)
_RE_HAN = re.compile(r"([一-鿕]+)", re.U)
_RE_ENG = re.compile(r"([a-zA-Z0-9]+(?:\.\d+)?%?)", re.U)
# ASCII multi-character entries of jieba's dict.txt [from memory — verify]
_ASCII_DICT_WORDS = ("AT&T", "C++", "c++", "C#", "c#")


def tokenize_en(text: str) -> List[str]:
    return _EN_INDEX_RE.findall(text.lower())


def _finalseg_cut(buf: str) -> List[str]:
    out: List[str] = []
    for blk in _RE_HAN.split(buf):
        if not blk:
            continue
        if _RE_HAN.match(blk):
            out.extend(list(blk))  # no dictionary / HMM tables: one char per token (inexact)
        else:
            out.extend(x for x in _RE_ENG.split(blk) if x)
    return out


def _cut_block(blk: str) -> List[str]:
    out: List[str] = []
    buf = ""
    i = 0
    n = len(blk)

    def flush():
        nonlocal buf
        if buf:
            if len(buf) == 1:
                out.append(buf)
            else:
                out.extend(_finalseg_cut(buf))
            buf = ""

    while i < n:
        hit = next((w for w in _ASCII_DICT_WORDS if blk.startswith(w, i)), None)
        if hit:
            flush()
            out.append(hit)
            i += len(hit)
        else:
            buf += blk[i]
            i += 1
    flush()
    return out


def jieba_cut_restated(sentence: str) -> List[str]:
    out: List[str] = []
    for blk in _RE_BLOCK.split(sentence):
        if not blk:
            continue
        if _RE_BLOCK.match(blk):
            out.extend(_cut_block(blk))
        else:
            for x in _RE_SKIP.split(blk):
                if _RE_SKIP.match(x):
                    out.append(x)
                else:
                    out.extend(list(x))
    return out


def jieba_cut(sentence: str) -> List[str]:
    """`list(jieba.cut(sentence))` — the wheel if present, else the restatement."""
    if HAVE_JIEBA:
        return list(_jieba.cut(sentence))
    return jieba_cut_restated(sentence)


_RE_ZH = re.compile(r"[一-鿿]")
_RE_LATIN = re.compile(r"[A-Za-z]")


def detect_lang(text: str) -> str:
    """legalrag/utils/lang.py:9-15."""
    if not text:
        return "zh"
    return "en" if len(_RE_LATIN.findall(text)) > len(_RE_ZH.findall(text)) else "zh"
