"""Tokenisers: index-side English regex (bm25_builder.py:18-19) and the
restatement of jieba.cut's default mode for non-Han text."""
import pytest

from legal_rag_amd import text


def test_index_tokenizer_lowercases_and_keeps_apostrophes():
    assert text.tokenize_en("The Seller's warranty; § 2-314(a).") == ["the", "seller's", "warranty", "2", "314", "a"]
    assert text.tokenize_en("") == []


@pytest.mark.parametrize("q,exp", [
    ("What is § 2-314?", ["What", " ", "is", " ", "§", " ", "2", "-", "314", "?"]),   # SURVEY.md §7 worked example
    ("rate of 3.5% p.a.", ["rate", " ", "of", " ", "3.5%", " ", "p", ".", "a", "."]),
    ("a--b", ["a", "--", "b"]),
    ("x", ["x"]),
    ("", []),
    ("tab\there\r\nnext", ["tab", "\t", "here", "\r\n", "next"]),
    ("C++ and AT&T", ["C++", " ", "and", " ", "AT&T"]),
    ("(a) buyer", ["(", "a", ")", " ", "buyer"]),
])
def test_jieba_restatement_ascii(q, exp):
    assert text.jieba_cut_restated(q) == exp


def test_query_tokens_are_not_lowercased():
    toks = text.jieba_cut("Short Titles")
    assert "Short" in toks and "short" not in toks  # capitalised words never hit the lower-cased en index


def test_detect_lang():
    assert text.detect_lang("") == "zh"
    assert text.detect_lang("What is a merchant?") == "en"
    assert text.detect_lang("认购书是否属于预约合同") == "zh"
    assert text.detect_lang("合同 contract law rules") == "en"
