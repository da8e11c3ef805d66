"""The oracle against its own committed known answers (tests/golden/gen_oracle_golden.py)."""
import numpy as np

from conftest import GOLDEN, load_golden
from oracle import bm25 as OB
from oracle import dense as OD
from oracle import maxsim as OM


def unit_rows(rng, n, d):
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X


def test_dense_golden():
    g = np.load(GOLDEN / "dense_flatip_golden.npz")
    rng = np.random.default_rng(0)
    X, Q = unit_rows(rng, 4096, 768), unit_rows(rng, 16, 768)
    s, i = OD.flatip_topk(X, Q, 10)
    assert np.array_equal(i, g["ids"]) and np.max(np.abs(s - g["scores"])) < 1e-5


def test_topk_ties_and_padding():
    s = np.array([[0.5, 0.9, 0.5, 0.9, 0.1]], dtype=np.float32)
    sc, idx = OD.topk_desc(s, 7)
    assert idx[0].tolist() == [1, 3, 0, 2, 4, -1, -1]
    ms, mi = OD.merge_topk([np.array([[0.9, 0.5]]), np.array([[0.9, 0.7]])], [np.array([[7, 9]]), np.array([[2, 8]])], 3)
    assert mi[0].tolist() == [2, 7, 8] and ms[0].tolist() == [0.9, 0.9, 0.7]


def test_bm25_toy_golden_and_hand_check():
    g = load_golden("bm25_toy.json")
    bm = OB.BM25Okapi([OB.tokenize_en(t) for t in g["docs"]])
    assert bm.idf == g["idf"] and bm.avgdl == g["avgdl"] and bm.doc_len == g["doc_len"]
    for case in g["queries"]:
        assert bm.get_scores(case["tokens"]).tolist() == case["scores"]
        assert [i for i, _ in OB.search(bm, case["tokens"], len(g["docs"]))] == case["order"]
    # hand check of one number: query ["goods"], doc 5 = "goods goods goods" (len 3, tf 3)
    import math
    N, df = 8, 6
    idf = math.log(N - df + 0.5) - math.log(df + 0.5)
    assert idf < 0  # -> floored to epsilon * average_idf
    idf = 0.25 * bm.average_idf
    expect = idf * (3 * 2.5 / (3 + 1.5 * (0.25 + 0.75 * 3 / bm.avgdl)))
    assert bm.get_scores(["goods"])[5] == expect
    # zero-score documents are returned, ties in ascending doc order (bm25_retriever.py:75)
    assert [i for i, _ in OB.search(bm, ["zzz"], 4)] == [0, 1, 2, 3]


# Published known answer of the upstream package: the README of rank_bm25 (0.2.x) indexes these
# three sentences split on blanks, queries "windy London" and prints
#     array([0.        , 0.93729472, 0.        ])
# — the only numeric BM25Okapi vector available offline (the wheel is absent; the reference's own
# tests hold none).  It exercises idf = ln(N-df+0.5) - ln(df+0.5), the
# epsilon floor (the idf of "is", df = 2 of 3, is negative and becomes 0.25 * mean idf) and
# the k1 / b length normalisation.
RANK_BM25_README_CORPUS = ["Hello there good man!", "It is quite windy in London", "How is the weather today?"]
RANK_BM25_README_QUERY = "windy London"
RANK_BM25_README_SCORES = [0.0, 0.93729472, 0.0]


def test_bm25_matches_rank_bm25_readme_example():
    from oracle import bm25 as OB
    ob = OB.BM25Okapi([d.split(" ") for d in RANK_BM25_README_CORPUS])
    got = ob.get_scores(RANK_BM25_README_QUERY.split(" "))
    assert [round(float(x), 8) for x in got] == RANK_BM25_README_SCORES
    assert ob.idf["is"] > 0 and abs(ob.idf["is"] - 0.25 * ob.average_idf) < 1e-15  # epsilon floor applied


def test_maxsim_golden():
    g = np.load(GOLDEN / "maxsim_golden.npz")
    rng = np.random.default_rng(42)
    lens = rng.integers(1, 221, size=64)
    lens[0], lens[-1] = 1, 220
    assert np.array_equal(lens, g["lens"])
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    D = unit_rows(rng, int(ptr[-1]), 128)
    Q = unit_rows(rng, 64, 128).reshape(2, 32, 128)
    assert np.allclose(OM.maxsim_scores(Q, D, ptr), g["scores"], atol=1e-12)
