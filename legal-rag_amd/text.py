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
Han runs cannot be segmented without dict.txt.  That case is never silent:
  * default: `jieba_cut` raises `ZhTokenizerUnavailable` when the text holds Han
    characters and neither jieba nor a registered tokenizer is available;
  * `register_tokenizer(fn, name)` plugs in an exact segmenter (a caller that
    has jieba elsewhere, or any jieba-compatible callable);
  * pre-tokenised entry points (`build_bm25_index(..., tokens=)`,
    `BM25Retriever.search(..., tokens=)`) bypass tokenisation altogether;
  * explicit opt-in to the inexact one-character-per-token stand-in:
    `cfg.retrieval.zh_tokenizer = "char"` (or LEGALRAG_ZH_TOKENIZER=char, or
    `mode="char"`): a WARNING is logged once, the index records tokenizer id
    "char" and every consumer reports `zh_exact: False`.
"""
from __future__ import annotations

import logging
import os
import re
from typing import Callable, List, Optional

logger = logging.getLogger(__name__)

try:  # pragma: no cover - not installed in the build container
    import jieba as _jieba
    HAVE_JIEBA = True
except Exception:  # noqa: BLE001
    _jieba = None
    HAVE_JIEBA = False


class ZhTokenizerUnavailable(RuntimeError):
    """Han text needs jieba's dictionary segmentation and none is available."""


_custom_cut: Optional[Callable[[str], List[str]]] = None
_custom_name: Optional[str] = None
_warned_char = False


def register_tokenizer(fn: Optional[Callable[[str], List[str]]], name: str = "custom") -> None:
    """Use `fn(sentence) -> tokens` wherever the reference calls jieba.cut (None removes it)."""
    global _custom_cut, _custom_name
    _custom_cut, _custom_name = fn, (name if fn is not None else None)


def zh_exact() -> bool:
    """True when Han text is segmented by jieba (or a registered exact tokenizer)."""
    return HAVE_JIEBA or _custom_cut is not None


def tokenizer_id(mode: Optional[str] = None) -> str:
    """Id recorded in bm25.pkl next to the index: which segmenter produced its tokens."""
    if _custom_cut is not None:
        return str(_custom_name)
    if HAVE_JIEBA:
        return "jieba"
    return "char" if resolve_mode(mode) == "char" else "jieba-restated-ascii"


def resolve_mode(mode: Optional[str]) -> str:
    m = (mode or os.environ.get("LEGALRAG_ZH_TOKENIZER") or "jieba").strip().lower()
    return "char" if m == "char" else "jieba"


def cfg_mode(cfg) -> Optional[str]:
    """`cfg.retrieval.zh_tokenizer` if the (duck-typed) config has it."""
    return getattr(getattr(cfg, "retrieval", None), "zh_tokenizer", None)


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


_ASCII_DICT_MARKS = frozenset("&+#")  # every entry of _ASCII_DICT_WORDS holds one of these


def _cut_block(blk: str) -> List[str]:
    if _ASCII_DICT_MARKS.isdisjoint(blk):  # no dictionary word can start anywhere in the block
        return [blk] if len(blk) == 1 else _finalseg_cut(blk)
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


def contains_han(sentence: str) -> bool:
    return _RE_HAN.search(sentence) is not None


def jieba_cut(sentence: str, mode: Optional[str] = None) -> List[str]:
    """`list(jieba.cut(sentence))`: a registered tokenizer, else the wheel, else — for text
    without Han characters only — the exact restatement.  Han text without a segmenter raises
    ZhTokenizerUnavailable unless the one-character stand-in was chosen explicitly (`mode` /
    LEGALRAG_ZH_TOKENIZER = "char"); that choice is logged once."""
    global _warned_char
    if _custom_cut is not None:
        return list(_custom_cut(sentence))
    if HAVE_JIEBA:
        return list(_jieba.cut(sentence))
    if not contains_han(sentence):
        return jieba_cut_restated(sentence)
    if resolve_mode(mode) != "char":
        raise ZhTokenizerUnavailable(
            "text contains Han characters but jieba is not importable: BM25 tokens would differ from the "
            "reference (bm25_builder.py:43, bm25_retriever.py:73). Install jieba, call "
            "legal_rag_amd.text.register_tokenizer(fn), pass pre-tokenised input (tokens=...), or opt in to the "
            "inexact one-character-per-token stand-in with cfg.retrieval.zh_tokenizer='char' / "
            "LEGALRAG_ZH_TOKENIZER=char")
    if not _warned_char:
        _warned_char = True
        logger.warning("[BM25] jieba is not importable: Han text is tokenised one character per token "
                       "(zh_tokenizer='char'). zh BM25 results differ from the reference; zh_exact=False")
    return jieba_cut_restated(sentence)


_RE_ZH = re.compile(r"[一-鿿]")
_RE_LATIN = re.compile(r"[A-Za-z]")


def detect_lang(text: str) -> str:
    """legalrag/utils/lang.py:9-15."""
    if not text:
        return "zh"
    return "en" if len(_RE_LATIN.findall(text)) > len(_RE_ZH.findall(text)) else "zh"
