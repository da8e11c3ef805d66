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


# ---- Han text is never tokenised silently by a stand-in (bm25_retriever.py:73, bm25_builder.py:43) ----
ZH = "当事人订立合同"


def test_han_without_segmenter_raises(monkeypatch):
    monkeypatch.setattr(text, "HAVE_JIEBA", False)
    monkeypatch.setattr(text, "_custom_cut", None)
    monkeypatch.delenv("LEGALRAG_ZH_TOKENIZER", raising=False)
    with pytest.raises(text.ZhTokenizerUnavailable):
        text.jieba_cut(ZH)
    assert text.jieba_cut("plain ASCII, no Han") == text.jieba_cut_restated("plain ASCII, no Han")  # exact restatement
    assert not text.zh_exact()
    assert text.tokenizer_id() == "jieba-restated-ascii"


def test_char_mode_is_explicit_and_logged(monkeypatch, caplog):
    monkeypatch.setattr(text, "HAVE_JIEBA", False)
    monkeypatch.setattr(text, "_custom_cut", None)
    monkeypatch.setattr(text, "_warned_char", False)
    with caplog.at_level("WARNING"):
        assert text.jieba_cut(ZH, "char") == list(ZH)
        text.jieba_cut(ZH, "char")
    assert sum("one character per token" in r.getMessage() for r in caplog.records) == 1  # once
    assert text.tokenizer_id("char") == "char"
    monkeypatch.setenv("LEGALRAG_ZH_TOKENIZER", "char")
    assert text.jieba_cut(ZH) == list(ZH)


def test_wheel_or_registered_tokenizer_wins(monkeypatch):
    class FakeJieba:
        @staticmethod
        def cut(s):
            return iter(["当事人", "订立", "合同"]) if s == ZH else iter([s])
    monkeypatch.setattr(text, "_custom_cut", None)
    monkeypatch.setattr(text, "HAVE_JIEBA", True)
    monkeypatch.setattr(text, "_jieba", FakeJieba, raising=False)
    assert text.jieba_cut(ZH) == ["当事人", "订立", "合同"]
    assert text.zh_exact() and text.tokenizer_id() == "jieba"
    monkeypatch.setattr(text, "HAVE_JIEBA", False)
    text.register_tokenizer(lambda s: s.split("立"), "mine")
    try:
        assert text.jieba_cut(ZH) == ["当事人订", "合同"]
        assert text.zh_exact() and text.tokenizer_id() == "mine"
    finally:
        text.register_tokenizer(None)
    assert not text.zh_exact()


def _zh_chunks():
    from legal_rag_amd.schemas import LawChunk
    return [LawChunk(id=f"z{i}", law_name="民法典", chapter="c", section="s", article_no=str(i), article_id=f"a{i}",
                     text=t, lang="zh", source="s.txt") for i, t in enumerate(["当事人订立合同", "合同的内容由当事人约定"])]


def test_bm25_builder_refuses_han_without_segmenter_and_records_tokenizer(tmp_path, monkeypatch):
    from types import SimpleNamespace
    from legal_rag_amd import artifacts
    from legal_rag_amd.retrieval.builders.bm25_builder import build_bm25_index
    monkeypatch.setattr(text, "HAVE_JIEBA", False)
    monkeypatch.setattr(text, "_custom_cut", None)
    monkeypatch.delenv("LEGALRAG_ZH_TOKENIZER", raising=False)
    chunks = _zh_chunks()
    cfg = SimpleNamespace(retrieval=SimpleNamespace(bm25_index_file=str(tmp_path / "bm25.pkl")))
    with pytest.raises(text.ZhTokenizerUnavailable):
        build_bm25_index(cfg, chunks)
    assert not (tmp_path / "bm25.pkl").exists()
    # explicit opt-in: built, and the pickle says so
    cfg.retrieval.zh_tokenizer = "char"
    build_bm25_index(cfg, chunks)
    bm, back = artifacts.read_bm25_pickle(tmp_path / "bm25.pkl")
    assert bm.__dict__["_tokenizer_id"] == "char" and [c.id for c in back] == ["z0", "z1"]
    assert set(bm.idf) == set("".join(c.text for c in chunks))
    # pre-tokenised documents: the exact path without jieba
    cfg.retrieval.zh_tokenizer = "jieba"
    build_bm25_index(cfg, chunks, tokens=[["当事人", "订立", "合同"], ["合同", "的", "内容", "由", "当事人", "约定"]],
                     tokenizer="jieba-offline")
    bm, _ = artifacts.read_bm25_pickle(tmp_path / "bm25.pkl")
    assert bm.__dict__["_tokenizer_id"] == "jieba-offline" and "当事人" in bm.idf and bm.doc_len == [3, 6]
    with pytest.raises(ValueError):
        build_bm25_index(cfg, chunks, tokens=[["x"]])


# ---- native batched tokeniser (csrc/tokenize.cpp): token for token what text.jieba_cut returns for non-Han text ----
NATIVE_CASES = ["What is § 2-314?", "rate of 3.5% p.a.", "a--b", "x", "", "tab\there\r\nnext", "C++ and AT&T",
                "(a) buyer", "1.2.3 a1.5b 50%.x  é—ü\u3000z", "c#c++x AT&T&", "AT&TC++C#", "a.b.c 1.%  2.5%% _x_ -- + #",
                "\r\n\r \n\x0b\x0c\x1c\x85\xa0\u2003\u2028end", "UPPER lower MiXeD 007", "§§ 9-102(a)(1)—“goods”",
                "trailing.", ".leading", "100%", "%", "a" * 300]


@pytest.mark.parametrize("q", NATIVE_CASES)
def test_native_tokenizer_equals_python_rule(q):
    from legal_rag_amd import _native
    assert _native.Tokenizer.cut(q) == text.jieba_cut_restated(q)


def test_native_tokenizer_on_the_ucc_queries_and_corpus_and_fuzz():
    """Every query of the 1 168-query UCC-en evaluation set, every chunk text of the corpus, and a seeded fuzz over
    an alphabet of letters, digits, the block punctuation, the dictionary marks, ASCII / Unicode whitespace and
    non-ASCII symbols: identical token lists."""
    import numpy as np
    from conftest import GOLDEN
    from legal_rag_amd import _native
    from legal_rag_amd.evaluation import synthetic_queries
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    chunks = load_chunks_from_dir(str(GOLDEN / "corpus"), "law_en.jsonl")
    qs = [q for q, _, _ in synthetic_queries(chunks, seed=0)]
    assert len(qs) == 1168
    for t in qs + [c.text for c in chunks[::7]]:
        assert _native.Tokenizer.cut(t) == text.jieba_cut_restated(t), t[:80]
    rng = np.random.default_rng(11)
    alphabet = list("abcXYZ0159") + list("+#&._%-") * 2 + list(" \t\n\r") + ["\r\n", "§", "é", "\u3000", "\xa0", "(", ")", ",", "C++", "AT&T", "c#"]
    for _ in range(3000):
        t = "".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=int(rng.integers(0, 40))))
        assert _native.Tokenizer.cut(t) == text.jieba_cut_restated(t), repr(t)
    assert _native.Tokenizer.cut("第四百九十五条 contract") is None  # Han text is never cut by the native rule


def test_native_encode_is_the_term_id_csr_of_the_python_path():
    import numpy as np
    from legal_rag_amd import _native
    from legal_rag_amd.bm25_model import BM25Okapi
    docs = [text.tokenize_en(t) for t in ("The buyer may reject goods.", "Merchant means a person; 3.5% rate", "c++ and AT&T § 2-314")]
    docs.append(["C++", "§", " ", "AT&T", "--"])  # tokens only the query-side rule produces
    bm = BM25Okapi(docs)
    tok = _native.Tokenizer(list(bm.vocab().keys()))
    qs = ["the buyer may reject", "The Buyer", "", "3.5% rate of C++ -- AT&T", "合同 buyer", "goods.  §", "   "]
    terms, q_ptr, hard = tok.encode(qs)
    assert hard.tolist() == [False, False, False, False, True, False, False]
    assert q_ptr[0] == 0 and q_ptr[-1] == len(terms)
    for i, q in enumerate(qs):
        got = terms[q_ptr[i]:q_ptr[i + 1]].tolist()
        assert got == ([] if hard[i] else bm.term_ids(text.jieba_cut_restated(q))), q
    assert (terms >= -1).all() and (terms < len(bm.vocab())).all() and (terms == -1).any()
    ids, ptr = _native.BM25Index.pack_queries([bm.term_ids(text.jieba_cut_restated(q)) for q in qs if not text.contains_han(q)])
    keep = [i for i in range(len(qs)) if not hard[i]]
    assert np.array_equal(np.concatenate([terms[q_ptr[i]:q_ptr[i + 1]] for i in keep]), ids[: ptr[-1]])
