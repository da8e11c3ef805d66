"""Parity of the HIP kernels (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): bit-exact doc-id ranks and fp64 scores for
BM25 and fusion; cosine / MaxSim scores within 1e-4 (fp32)."""
import copy
import math

import numpy as np
import pytest

from conftest import load_golden
from helpers import assert_hits_equal_mod_ties

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def nat():
    from legal_rag_amd import _native
    _native.load()
    assert _native.device_count() >= 1, "no GPU visible"
    assert _native.device_name(0).startswith("gfx950"), _native.device_name(0)
    return _native


def unit_rows(rng, n, d):
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X


def check_dense(nat, X, Q, k):
    from oracle import dense as OD
    idx = nat.DenseIndex(X)
    s, i = idx.search(Q, k)
    ref = OD.flatip_scores(X, Q).astype(np.float64)
    n = X.shape[0]
    kk = min(k, n)
    es, ei = OD.flatip_topk(X, Q, k)
    for b in range(Q.shape[0]):
        got_ids = i[b, :kk]
        assert len(set(got_ids.tolist())) == kk and got_ids.min() >= 0 and got_ids.max() < n
        # scores reported == exact score of the id reported, within 1e-4
        assert np.max(np.abs(s[b, :kk] - ref[b, got_ids])) <= TOL
        # sorted descending (by the kernel's own scores)
        assert np.all(np.diff(s[b, :kk]) <= 0)
        # same set as the oracle, except around near-ties at the k-th score
        kth = es[b, kk - 1]
        assert np.all(ref[b, got_ids] >= kth - TOL)
        clear = ref[b, ei[b, :kk]] > kth + TOL  # oracle hits clearly above the cut must be present
        assert set(ei[b, :kk][clear].tolist()) <= set(got_ids.tolist())
        # rank agreement wherever neighbouring oracle scores are separated by > tol
        gaps_ok = np.abs(np.diff(es[b, :kk])) > TOL
        same = got_ids == ei[b, :kk]
        sep = np.concatenate([[True], gaps_ok]) & np.concatenate([gaps_ok, [True]])
        assert np.all(same[sep]), (b, got_ids, ei[b, :kk])
        if k > n:
            assert np.all(i[b, n:] == -1)
            assert np.all(s[b, n:] == -np.finfo(np.float32).max)
    idx.close()
    return s, i


@pytest.mark.parametrize("n,d,nq,k", [
    (4096, 768, 16, 10),     # SURVEY §8c golden shape (3)
    (591, 768, 5, 10),       # UCC-en sized
    (591, 384, 3, 80),       # bge-small, eval depth
    (1260, 1024, 9, 30),     # zh sized, bge-m3 dim
    (7, 768, 2, 10),         # k > n: -1 padding
    (1, 768, 1, 1),
    (100000, 768, 8, 10),    # multi-block slabs, 8 queries per pass
    (50000, 256, 33, 128),   # ragged query groups, deep k
    (3000, 64, 4, 256),      # max k
    (5000, 384, 40, 80),     # batched fp32-MFMA path (nq >= 12), bge-small dim, ragged last query tile
    (20000, 1024, 33, 256),  # batched path, d = 1024 (2-wave blocks), max k, multi-slab top-k + merge
    (70000, 768, 64, 10),    # batched path, two full query tiles, several row slabs
    (33, 768, 12, 50),       # batched path, k > n and a single partial row tile
])
def test_dense_topk_matches_oracle(nat, n, d, nq, k):
    rng = np.random.default_rng(n * 31 + d + nq)
    X = unit_rows(rng, n, d)
    Q = unit_rows(rng, nq, d)
    check_dense(nat, X, Q, k)


def test_dense_fuzz_shapes(nat):
    """Seeded sweep over awkward shapes: both dense forms (GEMV for < 5 queries or d % 64 != 0,
    32-query MFMA tiles otherwise), partial row tiles, partial query tiles, k around n."""
    rng = np.random.default_rng(20261004)
    for it in range(40):
        d = int(rng.choice([4, 36, 64, 100, 128, 192, 320, 384, 512, 704, 768, 832, 1024]))
        n = int(rng.choice([1, 2, 31, 32, 33, 63, 65, 257, 591, 1023, 1260, 2500, 6000]))
        nq = int(rng.choice([1, 2, 4, 5, 7, 12, 31, 32, 33, 70]))
        k = int(rng.choice([1, 3, 10, 64, 80, 129, 256]))
        X = unit_rows(rng, n, d)
        Q = unit_rows(rng, nq, d)
        check_dense(nat, X, Q, k)


def test_dense_tile_kernel_long_batches(nat, monkeypatch):
    """The 32-query-tile kernel (dense_mfma.hip) serves 5-95 queries by default; AMDR_DENSE_PANEL=0
    pins it for long batches too (partial tiles on both sides, several row slabs)."""
    monkeypatch.setenv("AMDR_DENSE_PANEL", "0")
    rng = np.random.default_rng(77)
    for n, d, nq, k in [(591, 768, 2500, 10), (100, 64, 700, 5), (1999, 384, 1030, 20), (33, 1024, 300, 50),
                        (40, 832, 333, 7), (1, 128, 40, 3), (65, 192, 5, 80), (4097, 256, 95, 10)]:
        check_dense(nat, unit_rows(rng, n, d), unit_rows(rng, nq, d), k)


@pytest.mark.parametrize("parts", [None, "lo", "hi"])
def test_dense_panel_kernel_shapes(nat, parts, monkeypatch):
    """Long-batch form (dense_panel.hip: a block shares a panel of chunk rows through LDS): partial
    row blocks, partial query tiles, panels of 1..8 row blocks, parts of unequal size; the cut into
    parts pinned to its smallest and to a large count as well as the planner's choice."""
    monkeypatch.setenv("AMDR_DENSE_PANEL", "1")
    rng = np.random.default_rng(79)
    for n, d, nq, k in [(591, 768, 2500, 10), (100, 64, 700, 5), (1999, 384, 1030, 20), (33, 1024, 300, 50),
                        (40, 832, 333, 7), (1, 128, 140, 3), (65, 192, 97, 80), (4097, 256, 128, 10),
                        (1260, 768, 129, 10), (17, 64, 5, 4), (16, 64, 33, 16), (130, 320, 1000, 256)]:
        nb = (n + 15) // 16
        pmin = (nb + 7) // 8
        if parts == "lo":
            monkeypatch.setenv("AMDR_PANEL_PARTS", str(pmin))
        elif parts == "hi":
            monkeypatch.setenv("AMDR_PANEL_PARTS", str(min(nb, 2 * pmin + 3)))
        check_dense(nat, unit_rows(rng, n, d), unit_rows(rng, nq, d), k)


def test_dense_panel_persistent_grid_equals_one_block_per_tile(nat, monkeypatch):
    """More logical blocks than stay resident (dense_panel.hip launches occupancy x CUs persistent blocks that
    walk them): ragged last query tile, parts of unequal size, a grid that is not a multiple of the resident
    count; AMDR_PANEL_PERSIST=0 pins one block per logical block, =1 forces one resident block per CU (a longer
    walk).  Identical bits in all three, and the oracle's answer."""
    rng = np.random.default_rng(81)
    X, Q = unit_rows(rng, 591, 768), unit_rows(rng, 20011, 768)
    out = {}
    for flag in (None, "0", "1"):
        if flag is None:
            monkeypatch.delenv("AMDR_PANEL_PERSIST", raising=False)
        else:
            monkeypatch.setenv("AMDR_PANEL_PERSIST", flag)
        idx = nat.DenseIndex(X)
        out[flag] = idx.search(Q, 10)
        idx.close()
    for flag in ("0", "1"):
        assert np.array_equal(out[None][1], out[flag][1]) and np.array_equal(out[None][0], out[flag][0])
    monkeypatch.delenv("AMDR_PANEL_PERSIST", raising=False)
    sel = np.concatenate([np.arange(0, 300), np.arange(19800, 20011)])
    s, i = check_dense(nat, X, Q[sel], 10)
    assert np.array_equal(i, out[None][1][sel])


def test_dense_panel_agrees_bitwise_with_tiles(nat, monkeypatch):
    """The panel kernel and the 32x32-tile kernel feed the matrix pipe the same k order per
    (query, row): identical bits, identical ids."""
    rng = np.random.default_rng(80)
    X, Q = unit_rows(rng, 591, 768), unit_rows(rng, 1500, 768)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("AMDR_DENSE_PANEL", flag)
        idx = nat.DenseIndex(X)
        out[flag] = idx.search(Q, 10)
        idx.close()
    assert np.array_equal(out["0"][1], out["1"][1])
    assert np.array_equal(out["0"][0], out["1"][0])


def test_dense_two_level_topk_equals_full_score_matrix(nat, monkeypatch):
    """Large-scan form (dense.hip run_search_two_level): per-tile maxima -> candidate tiles -> exact re-scoring -> top-k.
    AMDR_DENSE_TWO_LEVEL=1 pins it on small matrices, =0 pins the full score matrix; both must return the same ids
    and the same score bits — ragged last tile, 5..95 queries, k = 1..256, and a matrix made of repeated rows (exact
    ties across tiles: lower id first) — and the oracle's answer."""
    rng = np.random.default_rng(91)
    cases = [(5000, 128, 32, 10), (4999, 768, 5, 1), (3333, 256, 95, 10), (2100, 64, 40, 32), (70000, 64, 33, 7),
             (20000, 64, 40, 80), (40000, 64, 33, 256)]
    for n, d, nq, k in cases:
        X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", flag)
            idx = nat.DenseIndex(X)
            out[flag] = idx.search(Q, k)
            idx.close()
        assert np.array_equal(out["1"][1], out["0"][1]), (n, d, nq, k)
        assert np.array_equal(out["1"][0], out["0"][0]), (n, d, nq, k)
    for _ in range(12):  # seeded sweep of shapes (the pinned form needs n >= 64 k rows)
        k = int(rng.choice([1, 3, 10, 17, 40, 100]))
        n = int(rng.integers(64 * k + 1, 64 * k + 30000))
        d = int(rng.choice([64, 128, 320]))
        nq = int(rng.integers(5, 96))
        X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", flag)
            idx = nat.DenseIndex(X)
            out[flag] = idx.search(Q, k)
            idx.close()
        assert np.array_equal(out["1"][1], out["0"][1]) and np.array_equal(out["1"][0], out["0"][0]), (n, d, nq, k)
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    check_dense(nat, unit_rows(rng, 4097, 192), unit_rows(rng, 17, 192), 10)
    base = unit_rows(rng, 700, 128)
    X = np.concatenate([base] * 5, axis=0)  # every row five times, 700 apart: ties in different tiles
    Q = unit_rows(rng, 12, 128)
    idx = nat.DenseIndex(X)
    s, i = idx.search(Q, 10)
    idx.close()
    ref = (base.astype(np.float64) @ Q.astype(np.float64).T).T
    for b in range(12):
        top = np.argsort(-ref[b], kind="stable")[:2]
        assert i[b, :5].tolist() == [int(top[0]) + 700 * j for j in range(5)]
        assert i[b, 5:10].tolist() == [int(top[1]) + 700 * j for j in range(5)]
        assert len(set(s[b, :5].tolist())) == 1 and len(set(s[b, 5:].tolist())) == 1



def test_dense_two_level_and_full_form_with_nan_scores(nat, monkeypatch):
    """NaN scores (a NaN in a chunk row) sort LAST in the full form (they still fill the tail of a top-k that has
    fewer than k real scores); the two-level form never returns them — a maximum drops NaN, so an all-NaN tile is no
    candidate — and pads with (id -1, -FLT_MAX), the convention for "fewer than k results".  The two forms agree on
    every real score, and differ only there; neither reads an unwritten list entry."""
    rng = np.random.default_rng(17)
    n, d, nq, k = 64 * 12 + 11, 64, 9, 10
    X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
    real = [3, 40, 100, 333, 500, 700]            # fewer than k rows keep a real score
    Xn = np.full_like(X, np.nan)
    Xn[real] = X[real]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", flag)
        idx = nat.DenseIndex(Xn)
        out[flag] = idx.search(Q, k)
        idx.close()
    exp = (X[real].astype(np.float64) @ Q.astype(np.float64).T).T
    for b in range(nq):
        order = [real[j] for j in np.argsort(-exp[b], kind="stable")]
        for flag in ("1", "0"):
            s, i = out[flag]
            assert i[b, :len(real)].tolist() == order, (flag, b)
            assert np.allclose(s[b, :len(real)], np.sort(exp[b])[::-1], atol=TOL)
        s2, i2 = out["1"]
        assert (i2[b, len(real):] == -1).all() and (s2[b, len(real):] == -np.finfo(np.float32).max).all()
        s0, i0 = out["0"]
        tail = i0[b, len(real):]
        assert np.isnan(s0[b, len(real):]).all() and (tail >= 0).all() and (tail < n).all()
        assert tail.tolist() == sorted(tail.tolist()) and not set(tail.tolist()) & set(real)
    # with at least k real scores the NaN rows change nothing: both forms, same bits
    Xm = X.copy()
    Xm[::7] = np.nan
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", flag)
        idx = nat.DenseIndex(Xm)
        res[flag] = idx.search(Q, k)
        idx.close()
    assert np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][0], res["0"][0])
    assert not np.isnan(res["1"][0]).any() and not (res["1"][1] % 7 == 0).any()


def test_dense_golden_fixture(nat):
    """Seeded fixture of SURVEY.md §8c(3): X[4096,768], Q[16,768], rng(0)."""
    g = np.load(str(__import__("conftest").GOLDEN / "dense_flatip_golden.npz"))
    rng = np.random.default_rng(0)
    X = unit_rows(rng, 4096, 768)
    Q = unit_rows(rng, 16, 768)
    s, i = check_dense(nat, X, Q, 10)
    assert np.array_equal(i, g["ids"])
    assert np.max(np.abs(s - g["scores"])) <= TOL


def test_dense_ties_lower_id_first(nat):
    rng = np.random.default_rng(3)
    base = unit_rows(rng, 50, 128)
    X = np.concatenate([base, base, base], axis=0)  # every row appears 3x -> exact ties
    Q = unit_rows(rng, 4, 128)
    idx = nat.DenseIndex(X)
    s, i = idx.search(Q, 9)
    for b in range(4):
        # hits come in triples of equal score with ascending ids r, r+50, r+100
        for t in range(3):
            tri = i[b, 3 * t:3 * t + 3]
            assert tri[1] == tri[0] + 50 and tri[2] == tri[0] + 100
            assert s[b, 3 * t] == s[b, 3 * t + 1] == s[b, 3 * t + 2]


def test_dense_pair_selector_mass_ties_and_agreement(nat, monkeypatch):
    """Short rows under a batch are ranked two queries per wave (scores_pair_topk_kernel).  Mass ties
    at the cut (every row identical: more than 32 survivors) must fall back to the general selector
    and still return the lowest ids; and both forms must agree bit for bit on ordinary data."""
    rng = np.random.default_rng(5)
    row = unit_rows(rng, 1, 64)
    X = np.repeat(row, 300, axis=0)
    Q = unit_rows(rng, 7, 64)
    idx = nat.DenseIndex(X)
    s, i = idx.search(Q, 10)
    assert all(i[b].tolist() == list(range(10)) for b in range(7))
    idx.close()
    X, Q = unit_rows(rng, 591, 768), unit_rows(rng, 333, 768)  # odd query count: the last wave holds one query
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_TOPK_PAIR", flag)
        idx = nat.DenseIndex(X)
        out[flag] = [idx.search(Q, k) for k in (1, 10, 32)]
        idx.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_dense_adversarial_ascending_scores(nat):
    """Rows ordered so every new row beats all previous ones (worst case for the
    threshold/staging selector): result must still be exact."""
    d = 64
    n = 20000
    q = np.zeros((1, d), dtype=np.float32)
    q[0, 0] = 1.0
    X = np.zeros((n, d), dtype=np.float32)
    X[:, 0] = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    X[:, 1] = 0.5
    idx = nat.DenseIndex(X)
    s, i = idx.search(q, 100)
    assert i[0].tolist() == list(range(n - 1, n - 101, -1))
    idx.close()


def test_dense_add_and_read_rows(nat):
    rng = np.random.default_rng(11)
    X = unit_rows(rng, 300, 768)
    idx = nat.DenseIndex(X[:100])
    idx.add(X[100:250])
    idx.add(X[250:])
    assert idx.ntotal == 300
    assert np.array_equal(idx.read_rows(90, 30), X[90:120])
    from oracle import dense as OD
    s, i = idx.search(X[[5, 170, 299]], 3)
    assert i[:, 0].tolist() == [5, 170, 299]
    es, ei = OD.flatip_topk(X, X[[5, 170, 299]], 3)
    assert np.array_equal(i, ei)


def test_dense_nontemporal_policy_changes_nothing_but_speed(nat, monkeypatch):
    """Matrices beyond the Infinity Cache are streamed with non-temporal loads (AMDR_DENSE_NT pins the
    choice): a cache policy, not arithmetic — GEMV scan, tile kernel and top-k passes give identical bits."""
    rng = np.random.default_rng(21)
    X = unit_rows(rng, 20000, 768)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("AMDR_DENSE_NT", flag)
        idx = nat.DenseIndex(X)
        out[flag] = [idx.search(unit_rows(np.random.default_rng(5), nq, 768), k) for nq, k in ((1, 10), (4, 80), (33, 10))]
        idx.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_handles_on_a_non_default_device(nat):
    """The C ABI takes a device ordinal; kernels that need the > 64 KiB dynamic-LDS opt-in set the
    attribute for the CURRENT device on every launch (a per-process "done" flag made a second device
    fail).  Needs two visible GPUs: the driver's 8-GPU node runs it, a one-GPU box skips."""
    if nat.device_count() < 2:
        pytest.skip("one visible GPU")
    rng = np.random.default_rng(12)
    X, Q = unit_rows(rng, 591, 768), unit_rows(rng, 200, 768)
    out = []
    for dev in (0, 1):
        idx = nat.DenseIndex(X, device=dev)
        out.append(idx.search(Q, 10))        # long batch: panel kernel (56 KiB LDS)
        out.append(idx.search(Q[:40], 10))   # short batch: 32x32 tile kernel (128 KiB LDS)
        idx.close()
    assert np.array_equal(out[0][1], out[2][1]) and np.array_equal(out[0][0], out[2][0])
    assert np.array_equal(out[1][1], out[3][1]) and np.array_equal(out[1][0], out[3][0])


def test_dense_errors(nat):
    with pytest.raises(nat.NativeError):
        nat.DenseIndex(np.zeros((4, 770), dtype=np.float32))  # dim not multiple of 4
    idx = nat.DenseIndex(np.zeros((4, 768), dtype=np.float32))
    with pytest.raises(nat.NativeError):
        idx.search(np.zeros((1, 768), dtype=np.float32), 0)
    with pytest.raises(nat.NativeError):
        idx.search(np.zeros((1, 768), dtype=np.float32), 257)
    with pytest.raises(ValueError):
        idx.search(np.zeros((1, 384), dtype=np.float32), 5)


# ---------------------------------------------------------------------------
def toy_corpus(rng, n_docs, vocab, max_len):
    words = [f"w{i}" for i in range(vocab)]
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    docs = []
    for _ in range(n_docs):
        ln = int(rng.integers(0, max_len + 1))
        docs.append([words[j] for j in rng.choice(vocab, size=ln, p=p)])
    if all(len(d) == 0 for d in docs):
        docs[0] = ["w0"]
    return docs, words


def bm25_pair(nat, docs):
    from oracle import bm25 as OB
    ob = OB.BM25Okapi(docs)
    csr = OB.to_csr(ob)
    gi = nat.BM25Index(csr["term_ptr"], csr["post_doc"], csr["post_tf"], csr["idf"], csr["doc_len"],
                       ob.avgdl, ob.k1, ob.b)
    return ob, csr, gi


@pytest.mark.parametrize("n_docs,vocab,max_len,k", [
    (591, 3926, 400, 10), (591, 3926, 400, 80), (12, 30, 8, 10), (5000, 500, 60, 10), (9000, 2000, 40, 256),
    (1, 5, 5, 3), (1260, 1181, 200, 10), (2048, 900, 30, 40), (2049, 900, 30, 10),
])
def test_bm25_bit_exact(nat, n_docs, vocab, max_len, k):
    from oracle import bm25 as OB
    rng = np.random.default_rng(n_docs + vocab + k)
    docs, words = toy_corpus(rng, n_docs, vocab, max_len)
    ob, csr, gi = bm25_pair(nat, docs)
    queries = []
    for qi in range(12):
        ln = int(rng.integers(0, 14))
        toks = [words[j] for j in rng.integers(0, min(vocab, 60), size=ln)]
        if qi % 3 == 0 and toks:
            toks += [toks[0], "UNKNOWN", toks[-1]]  # duplicates + out-of-vocabulary
        queries.append(toks)
    queries.append([])
    queries.append([words[j] for j in rng.integers(0, min(vocab, 60), size=100)] + ["UNKNOWN"] * 3)  # several token groups
    queries.append(["UNKNOWN"] * 40 + [words[0]])  # a first group without any known token
    tid = [[csr["vocab"].get(t, -1) for t in q] for q in queries]
    full = gi.get_scores(tid)
    s, i = gi.search(tid, k)
    for qn, q in enumerate(queries):
        ref = ob.get_scores(q)
        assert np.array_equal(full[qn], ref), f"query {qn}: scores differ"  # bit-exact fp64
        exp = OB.search(ob, q, k)
        kk = min(k, n_docs)
        assert i[qn, :kk].tolist() == [e[0] for e in exp]
        assert s[qn, :kk].tolist() == [e[1] for e in exp]
        assert np.all(i[qn, kk:] == -1)


def test_bm25_fp32_image_ties_fall_back_to_exact_order(nat):
    """The fast ranking picks candidates on fp32 images of the fp64 scores and must give up
    whenever two neighbours share an image but differ in fp64 (csrc/bm25.hip bm25_select_f32):
    hand-made idf values 1e-12 apart put the HIGHER score on the HIGHER doc id, so an order taken
    from the images alone (ties -> lower id) would be wrong."""
    n, V = 300, 40
    rng = np.random.default_rng(42)
    term_of_doc = rng.integers(0, V, size=n)           # every document holds exactly one term once
    order = np.argsort(term_of_doc, kind="stable")
    term_ptr = np.concatenate([[0], np.cumsum(np.bincount(term_of_doc, minlength=V))]).astype(np.int64)
    post_doc = order.astype(np.int32)
    post_tf = np.ones(n, dtype=np.int32)
    doc_len = np.full(n, 7, dtype=np.int32)
    idf = 1.0 + 1e-12 * np.arange(V, dtype=np.float64)  # all 40 images equal in fp32
    assert len(set(np.float32(idf).tolist())) == 1
    gi = nat.BM25Index(term_ptr, post_doc, post_tf, idf, doc_len, 7.0, 1.5, 0.75)
    queries = [list(range(V)), [5, 17, 39], [39], [0, 0, 3]]
    full = gi.get_scores(queries)
    for k in (10, 64, 80):
        s, i = gi.search(queries, k)
        for qn in range(len(queries)):
            exp = sorted(range(n), key=lambda d: full[qn][d], reverse=True)[:k]  # stable: ties -> lower id
            assert i[qn].tolist() == exp
            assert s[qn].tolist() == [full[qn][d] for d in exp]


def test_bm25_select_and_exact_rounds_agree(nat, monkeypatch):
    """AMDR_BM25_SELECT=0 pins the exact arg-max rounds: both rankings give identical bits on a
    Zipf corpus with queries that leave most documents at score 0 (mass ties at the cut)."""
    rng = np.random.default_rng(7)
    docs, words = toy_corpus(rng, 591, 3000, 300)
    ob, csr, gi = bm25_pair(nat, docs)
    queries = [[int(t) for t in rng.integers(0, 3000, size=int(rng.integers(0, 9)))] for _ in range(200)]
    queries += [[], [-1, -1], [2999]]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_BM25_SELECT", flag)
        out[flag] = [gi.search(queries, k) for k in (1, 10, 16, 40)]
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_bm25_mass_ties_at_the_cut(nat, monkeypatch):
    """More than 64 documents tie AT the k-th score: blocks of identical documents (equal non-zero fp64 scores),
    untouched documents at +0.0, and common-term (negative idf) documents below zero.  The fp32-image ranking
    takes everything above the cut plus the lowest ids of the tie (bm25_select_f32), the exact rounds are
    pinned by AMDR_BM25_SELECT=0; both must give the oracle's order."""
    from oracle import bm25 as OB
    docs = []
    for i in range(600):
        if i % 3 == 0:
            docs.append(["alpha", "beta"])            # 200 identical documents
        elif i % 3 == 1:
            docs.append(["gamma"] * (1 + i % 2))       # two more tie classes
        else:
            docs.append(["common", f"u{i}"])
    docs[17] = ["alpha", "alpha", "delta"]
    docs[401] = ["delta", "common"]
    ob, csr, gi = bm25_pair(nat, docs)
    queries = [["alpha"], ["alpha", "delta"], ["gamma"], ["common"], ["common", "alpha"], ["delta"], ["nothing"],
               ["u2", "u5", "alpha"], ["beta", "gamma", "common", "delta"]]
    tid = [[csr["vocab"].get(t, -1) for t in q] for q in queries]
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_BM25_SELECT", flag)
        for k in (1, 10, 16, 64, 80):
            s, i = gi.search(tid, k)
            for qn, q in enumerate(queries):
                exp = OB.search(ob, q, k)
                assert i[qn].tolist() == [e[0] for e in exp], (flag, k, q)
                assert s[qn].tolist() == [e[1] for e in exp], (flag, k, q)


def test_bm25_denormal_products_and_sums(nat):
    """The scatter adds with an LDS fp64 atomic (ds_add_f64): subnormal products, subnormal sums and sums that
    cross into the normal range must come out as numpy's fp64 gives them (no flush to zero)."""
    n, V = 100, 8
    term_ptr = np.arange(0, (V + 1) * n, n, dtype=np.int64)
    post_doc = np.tile(np.arange(n, dtype=np.int32), V)
    post_tf = np.ones(n * V, dtype=np.int32)
    doc_len = np.full(n, V, dtype=np.int32)
    idf = np.array([3e-310, -1e-310, 5e-324, 2.5e-308, -2.4e-308, 1e-309, 7e-311, 4e-320], dtype=np.float64)
    gi = nat.BM25Index(term_ptr, post_doc, post_tf, idf, doc_len, float(V), 1.5, 0.75)
    queries = [[0, 1, 2], [3, 4], [5, 6, 7, 2, 2], [3, 4, 0, 1], [3, 3, 3]]
    full = gi.get_scores(queries)
    w = np.float64(1.0) * (1.5 + 1) / (np.float64(1.0) + 1.5 * (1 - 0.75 + 0.75 * np.float64(V) / np.float64(V)))
    for q, row in zip(queries, full):
        s = np.float64(0.0)
        for t in q:
            s = s + idf[t] * w
        assert row.tolist() == [float(s)] * n


def test_bm25_toy_golden(nat):
    g = load_golden("bm25_toy.json")
    from oracle import bm25 as OB
    docs = [OB.tokenize_en(t) for t in g["docs"]]
    ob, csr, gi = bm25_pair(nat, docs)
    assert [float(x) for x in csr["idf"]] == [g["idf"][w] for w in csr["vocab"]]
    for case in g["queries"]:
        tid = [[csr["vocab"].get(t, -1) for t in case["tokens"]]]
        assert gi.get_scores(tid)[0].tolist() == case["scores"]
        s, i = gi.search(tid, len(docs))
        assert i[0].tolist() == case["order"]


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n_docs,nq,q_len,k", [(591, 3, 32, 10), (40, 2, 32, 80), (200, 2, 17, 5), (3, 1, 32, 10),
                                               # >= 8 queries: the blocked form (document tiles shared through LDS)
                                               (591, 20, 32, 10), (40, 9, 17, 80), (13, 8, 32, 5), (9, 33, 1, 3)])
def test_maxsim_matches_oracle(nat, n_docs, nq, q_len, k):
    from oracle import maxsim as OM
    rng = np.random.default_rng(n_docs + q_len)
    lens = rng.integers(1, 221, size=n_docs)
    lens[0] = 1
    lens[-1] = 220
    if n_docs > 5:
        lens[1:5] = [31, 32, 33, 64]
    doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    D = unit_rows(rng, int(doc_ptr[-1]), 128)
    Q = unit_rows(rng, nq * q_len, 128).reshape(nq, q_len, 128)
    # make some similarities strongly negative so padding rows would win if unmasked
    D[doc_ptr[0]] = -Q[0, 0]
    idx = nat.MaxSimIndex(D, doc_ptr)
    full = idx.scores(Q)
    ref = OM.maxsim_scores(Q, D, doc_ptr)
    assert np.max(np.abs(full - ref)) <= TOL
    s, i = idx.search(Q, k)
    es, ei = OM.maxsim_topk(Q, D, doc_ptr, k)
    kk = min(k, n_docs)
    for b in range(nq):
        got = i[b, :kk]
        assert len(set(got.tolist())) == kk
        assert np.max(np.abs(s[b, :kk] - ref[b, got])) <= TOL
        gaps_ok = np.abs(np.diff(es[b, :kk])) > 2 * TOL
        sep = np.concatenate([[True], gaps_ok]) & np.concatenate([gaps_ok, [True]])
        assert np.all((got == ei[b, :kk])[sep])
        assert np.all(i[b, kk:] == -1)


def test_maxsim_blocked_equals_per_query_bitwise(nat):
    """Both MaxSim kernels feed the same operands in the same k order to the same MFMA chain:
    a batch (blocked kernel) returns the bits of its queries run one by one."""
    rng = np.random.default_rng(4242)
    n_docs, nq, q_len = 77, 19, 32
    lens = rng.integers(1, 221, size=n_docs)
    lens[:6] = [1, 31, 32, 33, 64, 220]
    doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    D = unit_rows(rng, int(doc_ptr[-1]), 128)
    Q = unit_rows(rng, nq * q_len, 128).reshape(nq, q_len, 128)
    idx = nat.MaxSimIndex(D, doc_ptr)
    batch = idx.scores(Q)
    single = np.concatenate([idx.scores(Q[b:b + 1]) for b in range(nq)])
    assert np.array_equal(batch, single)


def test_maxsim_split_fp16_form_accuracy_scaling_and_pin(nat, monkeypatch):
    """The default MaxSim form runs the tile on v_mfma_f32_16x16x32_f16 with every operand split exactly into
    hi + lo / 2048 (22 significant bits, three MFMAs per block; csrc/maxsim.hip); AMDR_MAXSIM_F16X3=0 pins the
    fp32-input MFMA form.  Against the fp64 oracle the split form must stay FAR inside north_star's 1e-4 (a few 1e-6
    on scores of ~20 — no worse than the fp32-input form, whose 128-term fp32 chains round more often); the two forms
    agree to 2e-5; both kernels (blocked / per pair) return the same bits in either form; power-of-two scaling makes
    the result scale-free: a store 37.5 x larger and queries 1/1024 x smaller give the same relative errors, a store
    with wildly different token norms stays accurate relative to its largest token; a non-finite store falls back to
    the fp32-input form instead of poisoning a scale."""
    from oracle import maxsim as OM
    rng = np.random.default_rng(77)
    n_docs, nq, q_len = 150, 24, 32
    lens = rng.integers(1, 221, size=n_docs)
    lens[:6] = [1, 15, 16, 17, 32, 220]
    doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    D = unit_rows(rng, int(doc_ptr[-1]), 128)
    Q = unit_rows(rng, nq * q_len, 128).reshape(nq, q_len, 128)
    ref = OM.maxsim_scores(Q, D, doc_ptr)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AMDR_MAXSIM_F16X3", flag)
        idx = nat.MaxSimIndex(D, doc_ptr)
        out[flag] = idx.scores(Q)
        single = np.concatenate([idx.scores(Q[b:b + 1]) for b in range(0, nq, 5)])
        assert np.array_equal(out[flag][::5], single), flag  # blocked == per-pair kernel, bit for bit
        idx.close()
    e_half, e_f32 = np.max(np.abs(out["1"] - ref)), np.max(np.abs(out["0"] - ref))
    assert e_half <= 1e-5 and e_f32 <= 2e-5, (e_half, e_f32)
    assert not np.array_equal(out["1"], out["0"])  # the pin really selects another kernel
    assert np.max(np.abs(out["1"] - out["0"])) <= 2e-5
    monkeypatch.setenv("AMDR_MAXSIM_F16X3", "1")
    # scale-free: the same data, store x 37.5 (not a power of two), queries / 1024
    idx = nat.MaxSimIndex((D * np.float32(37.5)).astype(np.float32), doc_ptr)
    got = idx.scores((Q / np.float32(1024)).astype(np.float32))
    idx.close()
    ref_s = OM.maxsim_scores((Q / np.float32(1024)).astype(np.float32), (D * np.float32(37.5)).astype(np.float32), doc_ptr)
    assert np.max(np.abs(got - ref_s)) <= 1e-5 * 37.5 / 1024
    # token norms spread over 2^-10 .. 1: error stays relative to the LARGEST token (fixed-point-like), inside 1e-4
    w = np.exp2(-rng.integers(0, 11, size=D.shape[0])).astype(np.float32)[:, None]
    idx = nat.MaxSimIndex(D * w, doc_ptr)
    got = idx.scores(Q)
    idx.close()
    assert np.max(np.abs(got - OM.maxsim_scores(Q, D * w, doc_ptr))) <= 1e-5
    # a query block with an all-zero query and a tiny one
    Qz = Q.copy()
    Qz[3] = 0.0
    Qz[4] *= np.float32(1e-30)
    idx = nat.MaxSimIndex(D, doc_ptr)
    got = idx.scores(Qz)
    assert np.all(got[3] == 0.0) and np.allclose(got[4], ref[4] * 1e-30, rtol=1e-5, atol=0)
    assert np.max(np.abs(got[5:] - ref[5:])) <= 1e-5
    idx.close()
    # a store holding an infinity: no scale can be taken from it -> fp32-input form (results as before)
    Dn = D.copy()
    Dn[7, 3] = np.inf
    idx = nat.MaxSimIndex(Dn, doc_ptr)
    got = idx.scores(Q[:8])
    idx.close()
    bad = int(np.searchsorted(doc_ptr, 7, side="right") - 1)
    ok = np.ones(n_docs, bool)
    ok[bad] = False
    assert np.max(np.abs(got[:, ok] - ref[:8, ok])) <= 2e-5 and not np.isfinite(got[:, bad]).all()


def test_maxsim_two_pass_topk_equals_one_pass(nat, monkeypatch):
    """Batched `search` on the split-fp16 form first scores every document with the hi parts only, then re-scores with
    the full form the documents whose first-pass score is within the proven bound of the k-th best (csrc/maxsim.hip,
    "two-pass top-k"); AMDR_MAXSIM_TWOPASS=0 pins the one-pass form.  Same ids and the same score BITS — ragged
    documents, several k, a corpus of near-duplicates (candidate list overflow -> every document re-scored), exact
    duplicates across the cut (ties -> lower id), fewer documents than 4 k (one pass inside), wide token norms."""
    from oracle import maxsim as OM
    rng = np.random.default_rng(2026)

    def run(D, doc_ptr, Q, k):
        out = {}
        # "1": two passes, the candidates re-scored by document (round 4: a block = one document x 8 of its queries);
        # "1r3": two passes, one wave per candidate pair (round 3); "0": one pass
        # "1w": as "1" with the round-4 shortcuts off — every block of pass 1 splits its own queries, the final top-k
        # ranks whole re-scored rows
        for flag, rescore, short in (("1", "1", "1"), ("1r3", "0", "1"), ("0", "1", "1"), ("1w", "1", "0")):
            monkeypatch.setenv("AMDR_MAXSIM_TWOPASS", flag[0])
            monkeypatch.setenv("AMDR_MAXSIM_RESCORE", rescore)
            monkeypatch.setenv("AMDR_MAXSIM_PRESPLIT", short)
            monkeypatch.setenv("AMDR_MAXSIM_FINAL", short)
            idx = nat.MaxSimIndex(D, doc_ptr)
            out[flag] = idx.search(Q, k)
            idx.close()
        # the re-scoring pass's ring depth / blocks per CU (default 2 stages, grid of 8 blocks per CU): the persistent
        # form of the start of the round (4 stages, 2 blocks per CU) and one block per CU (every block walks many items)
        for flag, ring, blocks in (("1g4", "4", "2"), ("1g3", "3", "1")):
            monkeypatch.setenv("AMDR_MAXSIM_TWOPASS", "1")
            monkeypatch.setenv("AMDR_MAXSIM_RESCORE_RING", ring)
            monkeypatch.setenv("AMDR_MAXSIM_RESCORE_BLOCKS", blocks)
            idx = nat.MaxSimIndex(D, doc_ptr)
            out[flag] = idx.search(Q, k)
            idx.close()
        for name in ("AMDR_MAXSIM_RESCORE", "AMDR_MAXSIM_PRESPLIT", "AMDR_MAXSIM_FINAL", "AMDR_MAXSIM_RESCORE_RING",
                     "AMDR_MAXSIM_RESCORE_BLOCKS"):
            monkeypatch.delenv(name)
        for other in ("0", "1r3", "1w", "1g4", "1g3"):
            assert np.array_equal(out["1"][1], out[other][1]), (k, other)
            assert np.array_equal(out["1"][0], out[other][0]), (k, other)
        return out["1"]

    for n_docs, nq, q_len, ks in ((591, 24, 32, (1, 10, 80)), (130, 9, 17, (5, 32)), (1300, 16, 32, (10,)), (50, 8, 32, (10, 13))):
        lens = rng.integers(1, 221, size=n_docs)
        lens[:4] = [1, 63, 64, 65]
        doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        D = unit_rows(rng, int(doc_ptr[-1]), 128)
        Q = unit_rows(rng, nq * q_len, 128).reshape(nq, q_len, 128)
        for k in ks:
            s, i = run(D, doc_ptr, Q, k)
            ref = OM.maxsim_scores(Q, D, doc_ptr)
            kk = min(k, n_docs)
            for b in range(nq):
                assert np.max(np.abs(s[b, :kk] - ref[b, i[b, :kk]])) <= 1e-5
                assert np.all(ref[b, i[b, :kk]] >= np.sort(ref[b])[::-1][kk - 1] - 1e-5)
    # near-duplicates: 200 of 300 documents differ in the 4th decimal -> far more than `cap` candidates -> overflow path;
    # exact duplicates straddling the cut -> lower id first, in both forms
    base = unit_rows(rng, 40, 128)
    docs = [base + 1e-4 * rng.standard_normal(base.shape).astype(np.float32) for _ in range(200)]
    docs += [unit_rows(rng, int(rng.integers(5, 80)), 128) for _ in range(100)]
    docs[7] = docs[3].copy()
    docs[150] = docs[3].copy()
    D = np.concatenate(docs).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    doc_ptr = np.concatenate([[0], np.cumsum([len(x) for x in docs])]).astype(np.int64)
    Q = np.stack([np.concatenate([base[:20] + 0.05 * rng.standard_normal((20, 128)).astype(np.float32),
                                  unit_rows(rng, 12, 128)]) for _ in range(8)]).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=2, keepdims=True)
    s, i = run(D, doc_ptr, Q, 10)
    for b in range(8):
        pos = {int(d): j for j, d in enumerate(i[b])}
        if 3 in pos and 7 in pos and 150 in pos:
            assert pos[3] < pos[7] < pos[150] and s[b, pos[3]] == s[b, pos[7]] == s[b, pos[150]]
    # token norms over three orders of magnitude, a scaled store and tiny queries: the bound follows the norms
    w = np.exp2(-rng.integers(0, 10, size=D.shape[0])).astype(np.float32)[:, None]
    run((D * w * np.float32(12.5)).astype(np.float32), doc_ptr, (Q * np.float32(3e-3)).astype(np.float32), 10)


def test_maxsim_two_pass_on_the_ucc_token_store_vs_oracle(nat, monkeypatch):
    """The two-pass top-k (hi-only first pass, candidates re-scored by document in the full split-fp16 form) against
    oracle/maxsim.py (fp64 definition of sum_i max_j q_i . d_j) on the UCC-en token store itself — 591 documents, 95 k
    stand-in token vectors, 96 queries of the evaluation set — at k = 10 (serving) and k = 80 (the reference's
    evaluation depth): reported score == oracle score of the reported id within 1e-4 (north_star's bar), the oracle's hits
    clearly above the cut all present, ranks equal wherever the oracle's neighbouring scores are separated by > 1e-4."""
    from pathlib import Path
    from legal_rag_amd.encoders import HashingTokenEmbedder
    from legal_rag_amd.evaluation import synthetic_queries
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir
    from oracle import maxsim as OM
    chunks = load_chunks_from_dir(str(Path(__file__).resolve().parent / "golden" / "corpus"), "law_en.jsonl")
    te = HashingTokenEmbedder()
    mats = [te.encode_doc(c.text.strip()) for c in chunks]
    D = np.concatenate(mats, axis=0).astype(np.float32)
    doc_ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
    qs = synthetic_queries(chunks, seed=0)[::12][:96]
    Q = np.stack([te.encode_query(q.strip()) for q, _, _ in qs]).astype(np.float32)
    ref = OM.maxsim_scores(Q, D, doc_ptr)
    for rescore in ("1", "0"):
        monkeypatch.setenv("AMDR_MAXSIM_RESCORE", rescore)
        idx = nat.MaxSimIndex(D, doc_ptr)
        assert "two-pass" in idx.plan_info(len(Q)), idx.plan_info(len(Q))
        for k in (10, 80):
            s, i = idx.search(Q, k)
            for b in range(len(Q)):
                order = np.lexsort((np.arange(ref.shape[1]), -ref[b]))
                es, ei = ref[b, order[:k]], order[:k]
                assert len(set(i[b].tolist())) == k
                assert np.max(np.abs(s[b] - ref[b, i[b]])) <= TOL, (k, b)
                assert np.all(ref[b, i[b]] >= es[-1] - TOL)
                clear = es > es[-1] + TOL
                assert set(ei[clear].tolist()) <= set(i[b].tolist())
                gaps_ok = np.abs(np.diff(es)) > TOL
                sep = np.concatenate([[True], gaps_ok]) & np.concatenate([gaps_ok, [True]])
                assert np.all((i[b] == ei)[sep]), (k, b)
        idx.close()


def test_maxsim_fuzz_vs_oracle(nat):
    """Seeded sweep of both MaxSim forms (per-pair for < 8 queries, blocked otherwise): ragged
    document lengths around the 32-token tile, document counts around the 8-document group,
    short queries, batches that do not fill the last block."""
    from oracle import maxsim as OM
    rng = np.random.default_rng(20261005)
    for it in range(14):
        n_docs = int(rng.choice([1, 7, 8, 9, 16, 17, 63, 130]))
        nq = int(rng.choice([1, 3, 7, 8, 9, 15, 16, 31]))
        q_len = int(rng.choice([1, 5, 31, 32]))
        lens = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 220], size=n_docs)
        doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        D = unit_rows(rng, int(doc_ptr[-1]), 128)
        Q = unit_rows(rng, nq * q_len, 128).reshape(nq, q_len, 128)
        idx = nat.MaxSimIndex(D, doc_ptr)
        full = idx.scores(Q)
        ref = OM.maxsim_scores(Q, D, doc_ptr)
        assert full.shape == ref.shape and np.max(np.abs(full - ref)) <= TOL, (n_docs, nq, q_len)
        k = int(rng.choice([1, 5, 40]))
        s, i = idx.search(Q, k)
        kk = min(k, n_docs)
        for b in range(nq):
            got = i[b, :kk]
            assert len(set(got.tolist())) == kk and np.all(i[b, kk:] == -1)
            assert np.max(np.abs(s[b, :kk] - ref[b, got])) <= TOL
            assert np.all(ref[b, got] >= np.sort(ref[b])[::-1][kk - 1] - 2 * TOL)


def test_maxsim_rejects_empty_doc(nat):
    with pytest.raises(nat.NativeError):
        nat.MaxSimIndex(np.zeros((4, 128), np.float32), np.array([0, 2, 2, 4]))
    with pytest.raises(nat.NativeError):
        nat.MaxSimIndex(np.zeros((4, 64), np.float32), np.array([0, 4]))


# ---------------------------------------------------------------------------
FUSE = load_golden("fusion_golden.json")["cases"]
SEARCH = load_golden("search_golden.json")["cases"]


def _arr(case, ch, dtype):
    rows = sorted(case[ch], key=lambda p: -p[1])
    ids = np.array([[p[0] for p in rows]], dtype=np.int64).reshape(1, -1)
    sc = np.array([[p[1] for p in rows]], dtype=dtype).reshape(1, -1)
    return ids, sc


def _params(nat, kn, min_final=-math.inf):
    return nat.make_fuse_params(
        method=kn.get("fusion_method", "rrf_norm_blend"), rrf_k=kn.get("rrf_k", 60), alpha=kn.get("rrf_alpha", 0.5),
        w_dense=kn.get("dense_weight", 0.6), w_bm25=kn.get("bm25_weight", 0.4),
        w_colbert=kn.get("colbert_weight", 0.35), min_final_score=min_final)


def _hits_from_native(nat, ids, vals, mask, count, kn, *, rer=None, beta=None):
    names = ("dense", "bm25", "colbert")
    out = []
    for r in range(int(count)):
        v = vals[r]
        contrib = {n: float(v[6 + c]) for c, n in enumerate(names)}
        members = [n for c, n in enumerate(names) if mask[r] & (1 << c)]
        members.sort(key=lambda c: (contrib[c], c), reverse=True)
        sb = {
            "fusion_method": str(kn.get("fusion_method", "rrf_norm_blend")).lower(), "rrf_k": int(kn.get("rrf_k", 60)),
            "alpha": float(kn.get("rrf_alpha", 0.5)),
            "channel_weights": {"dense": float(kn.get("dense_weight", 0.6)), "bm25": float(kn.get("bm25_weight", 0.4)),
                                "colbert": float(kn.get("colbert_weight", 0.35))},
            "channel": members, "channel_contrib": contrib, "rrf_norm": float(v[1]), "weighted_sum": float(v[2]),
            "dense_norm": float(v[3]), "bm25_norm": float(v[4]), "colbert_norm": float(v[5]),
        }
        src = "retriever"
        if rer is not None and not math.isnan(rer[r, 0]):
            sb.update({"rerank_raw": float(rer[r, 0]), "rerank_norm": float(rer[r, 1]), "rerank_beta": beta})
            src = "rerank"
        out.append({"id": f"src.txt::{int(ids[r])}", "score": float(v[0]), "rank": r + 1, "source": src,
                    "breakdown": sb})
    return out


@pytest.mark.parametrize("case", FUSE, ids=[c["name"] for c in FUSE])
def test_fuse_kernel_bit_exact_vs_reference_vectors(nat, case):
    kn = case["knobs"]
    d = _arr(case, "dense", np.float32)
    b = _arr(case, "bm25", np.float64)
    c = _arr(case, "colbert", np.float32)
    ids, vals, mask, count = nat.fuse(_params(nat, kn), 1, d, b, c)
    got = _hits_from_native(nat, ids[0], vals[0], mask[0], count[0], kn)
    assert_hits_equal_mod_ties(got, case["expected"])


@pytest.mark.parametrize("case", SEARCH, ids=[c["name"] for c in SEARCH])
def test_fuse_filter_rerank_pipeline_vs_reference_vectors(nat, case):
    """fuse -> min_final filter -> rerank blend -> top_k through the C ABI,
    against vectors produced by the reference's HybridRetriever.search()."""
    from oracle import fusion as F
    kn = {k: v for k, v in case["knobs"].items() if k != "enable_graph"}
    full = dict(F.DEFAULTS)
    full.update(kn)
    top_k = case["top_k"]
    eff = F.eff_top_k(top_k, full["top_k"])
    cut = copy.deepcopy(case)
    for ch in ("dense", "bm25", "colbert"):
        cut[ch] = sorted(case[ch], key=lambda p: -p[1])[:eff]
    d = _arr(cut, "dense", np.float32)
    b = _arr(cut, "bm25", np.float64)
    c = _arr(cut, "colbert", np.float32)
    ids, vals, mask, count = nat.fuse(_params(nat, full, full["min_final_score"]), 1, d, b, c)
    rer = None
    if full["enable_rerank"] and count[0] > 0:
        n = min(int(full["rerank_top_n"]), int(count[0]))
        raw = np.array([[case["ce_raw_by_id"][f"src.txt::{int(i)}"] for i in ids[0, :n]]], dtype=np.float64)
        rer = nat.rerank_blend(count, ids, vals, mask, raw, float(full["rerank_beta"]))[0]
    got = _hits_from_native(nat, ids[0], vals[0], mask[0], count[0], full, rer=rer, beta=float(full["rerank_beta"]))
    assert_hits_equal_mod_ties(got[:top_k], case["expected"])


def test_fuse_and_rerank_reproduce_the_notebook_known_answers(nat):
    """The only fusion numbers the reference itself holds (notebooks/03_Retrieval_Performance_Evaluation.ipynb,
    query "认购书或者订购书等是否属于预约合同？"): article 495 is rank 1 in all three channels (:1068 BM25 37.36,
    :1188 dense 0.65, ColBERT 22.08), its fused score prints 1.18 (:1257) and its score after the rerank blend 1.11
    (:1529).  With the defaults (config.py:92-93,100,121-122: weights 0.6 / 0.4 / 0.35, rrf_alpha 0.5,
    rerank_beta 0.35) the formula gives alpha*1 + (1-alpha)*(0.6+0.4+0.35) = 1.175 and
    0.65*1.175 + 0.35*1 = 1.11375 — through amdr_fuse and amdr_rerank_blend, to the bit.  The channel lists are the
    notebook's printed top-5 (article ids and 2-digit scores); the hits it prints below 495 depend on ranks 6-10,
    which it does not show."""
    dense = [(495, 0.65), (491, 0.53), (471, 0.5299), (493, 0.51), (483, 0.5099)]
    bm25 = [(495, 37.36), (1134, 8.09), (501, 7.53), (250, 7.47), (254, 7.03)]
    colbert = [(495, 22.08), (502, 19.97), (493, 19.92), (888, 19.88), (984, 19.84)]

    def arr(pairs, dt):
        return (np.array([[i for i, _ in pairs]], dtype=np.int64), np.array([[s for _, s in pairs]], dtype=dt))
    params = nat.make_fuse_params(method="rrf_norm_blend", rrf_k=60, alpha=0.5, w_dense=0.6, w_bm25=0.4, w_colbert=0.35,
                                  min_final_score=0.2)
    ids, vals, mask, count = nat.fuse(params, 1, arr(dense, np.float32), arr(bm25, np.float64), arr(colbert, np.float32))
    assert ids[0, 0] == 495 and mask[0, 0] == 7
    assert vals[0, 0, nat.FV["rrf_norm"]] == 1.0 and vals[0, 0, nat.FV["weighted_sum"]] == 0.6 + 0.4 + 0.35
    assert vals[0, 0, nat.FV["score"]] == 0.5 * 1.0 + (1 - 0.5) * (0.6 + 0.4 + 0.35) == 1.175
    assert f"{vals[0, 0, 0]:.2f}" == "1.18"  # what the notebook prints
    # the hits the notebook lists under it all sit in dense AND colbert or dense only: 493 is in both here
    by_id = {int(i): r for r, i in enumerate(ids[0, :count[0]])}
    assert mask[0, by_id[493]] == 0b101
    n = int(count[0])
    raw = np.array([[0.99] + [0.5 - 0.01 * j for j in range(n - 1)]], dtype=np.float64)  # CE prefers 495 as well
    rr = nat.rerank_blend(count, ids, vals, mask, raw, 0.35)
    assert ids[0, 0] == 495 and rr[0, 0, 1] == 1.0
    assert vals[0, 0, 0] == (1 - 0.35) * 1.175 + 0.35 * 1.0
    assert abs(vals[0, 0, 0] - 1.11375) < 1e-15 and f"{vals[0, 0, 0]:.2f}" == "1.11"


def test_dense_search_fuse_equals_the_two_launches(nat, monkeypatch):
    """amdr_dense_search_fuse_device == amdr_dense_search_device + amdr_fuse_device(dense, bm25), bit for bit: the
    one-kernel form (<= 1 024 rows, k + kb <= 32: dense_select_fuse_kernel), the shapes that run the two launches
    inside the call, mass ties at the selector's cut (its staged fallback), id maps, an odd batch, no BM25 list."""
    import torch
    from legal_rag_amd.retrieval.engine import HybridEngine
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(123)
    params = nat.make_fuse_params(min_final_score=0.2)
    cases = [(591, 768, 130, 10, 10), (591, 768, 37, 10, 10), (37, 64, 5, 10, 10), (1024, 128, 33, 16, 16), (1, 64, 6, 10, 10),
             (1300, 64, 40, 10, 10), (591, 64, 257, 20, 20), (20000, 64, 9, 10, 10), (200, 64, 3, 10, 10), (640, 320, 96, 1, 31),
             (591, 128, 64, 10, 0)]
    for n, d, nq, k, kb in cases:
        X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
        if n == 591 and d == 768:  # mass ties: blocks of identical rows -> more than 32 keys at the cut for some queries
            X[100:180] = X[100]
        dense = nat.DenseIndex(X, device=0)
        eng = HybridEngine(dense, None, None, device=0)
        q_emb = torch.from_numpy(Q).to(dev)
        bs = torch.from_numpy(np.sort(rng.random((nq, max(kb, 1))) * 30.0, axis=1)[:, ::-1].copy()).to(dev)
        bi = torch.from_numpy(np.stack([rng.permutation(max(n, kb + 1))[: max(kb, 1)] for _ in range(nq)]).astype(np.int64)).to(dev)
        if kb:
            bi[0, kb - 2:] = -1  # a padded BM25 list
        maps = (torch.from_numpy(rng.permutation(n + 40)[:n].astype(np.int64)).to(dev),
                torch.from_numpy(rng.permutation(n + 40).astype(np.int64)).to(dev)) if n == 200 else (None, None)
        eng.maps = (maps[0], maps[1], None)
        b = (bs[:, :kb].contiguous(), bi[:, :kb].contiguous()) if kb else None
        out = {}
        for flag in ("1", "0", "sep"):
            if flag == "sep":  # the two C-ABI calls made separately
                dch = eng.dense_topk(q_emb, k)
                res = eng.fuse(params, nq, dch, b, None)
            else:
                monkeypatch.setenv("AMDR_DENSE_FUSE", flag)
                if kb:
                    dch, res = eng.dense_topk_fuse(params, q_emb, k, b)
                else:
                    z = torch.zeros((nq, 0), dtype=torch.float64, device=dev), torch.zeros((nq, 0), dtype=torch.int64, device=dev)
                    dch, res = eng.dense_topk_fuse(params, q_emb, k, z)
            torch.cuda.synchronize()
            out[flag] = [t.cpu().numpy().copy() for t in (dch[0], dch[1], res.ids, res.vals, res.mask, res.count)]
        for flag in ("0", "sep"):
            for a, e in zip(out["1"], out[flag]):
                assert a.shape == e.shape and np.array_equal(a, e, equal_nan=True), (n, d, nq, k, kb, flag)
        es, ei = __import__("oracle.dense", fromlist=["x"]).flatip_topk(X, Q, k)
        assert np.array_equal(out["1"][1], ei) or n == 591  # (tied rows: any order of the oracle's argsort is not pinned)
        dense.close()


def test_fuse_batched_equals_single(nat):
    rng = np.random.default_rng(9)
    nq, k = 37, 10
    di = np.stack([rng.choice(200, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    bi = np.stack([rng.choice(200, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    ds = -np.sort(-rng.uniform(0, 1, size=(nq, k)).astype(np.float32), axis=1)
    bs = -np.sort(-rng.uniform(0, 30, size=(nq, k)), axis=1)
    bi[3, 6:] = -1  # ragged channel
    p = nat.make_fuse_params(min_final_score=0.2)
    I, V, M, Cn = nat.fuse(p, nq, (di, ds), (bi, bs), None)
    for q in range(nq):
        i1, v1, m1, c1 = nat.fuse(p, 1, (di[q:q + 1], ds[q:q + 1]), (bi[q:q + 1], bs[q:q + 1]), None)
        assert np.array_equal(I[q], i1[0]) and np.array_equal(V[q], v1[0]) and np.array_equal(M[q], m1[0])
        assert Cn[q] == c1[0]


def test_fuse_fuzz_vs_oracle(nat):
    """Random channel lists (ragged depths up to 256, overlapping ids, random knobs, all four
    methods) through amdr_fuse vs the reference-pinned oracle: bit-exact, incl. the filter count."""
    from oracle import fusion as F
    rng = np.random.default_rng(77)
    methods = ["rrf_norm_blend", "rrf", "wrrf", "weighted_sum"]
    for it in range(120):
        pool = int(rng.choice([3, 12, 40, 300, 900]))
        ks = [int(rng.choice([0, 1, 5, 10, 80, 256])) for _ in range(3)]
        if sum(ks) == 0:
            ks[0] = 4
        chans = []
        for c, kk in enumerate(ks):
            kk = min(kk, pool)
            ids = rng.choice(pool, size=kk, replace=False).astype(np.int64)
            sc = np.sort(rng.uniform(-1, 40, size=kk))[::-1].copy()
            if c != 1:
                sc = sc.astype(np.float32).astype(np.float64)
            if it % 7 == 0 and kk > 2:
                sc[:] = sc[0]  # flat channel -> minmax degenerates to zeros
            chans.append((ids, sc))
        kn = {"fusion_method": methods[it % 4], "rrf_k": int(rng.choice([1, 10, 60])), "rrf_alpha": float(rng.uniform(0, 1)),
              "dense_weight": float(rng.uniform(0, 1)), "bm25_weight": float(rng.uniform(0, 1)),
              "colbert_weight": float(rng.uniform(0, 1))}
        min_final = float(rng.choice([-np.inf, 0.0, 0.2, 0.5]))
        args = [(i[None, :], s[None, :]) if len(i) else None for i, s in chans]
        ids, vals, mask, count = nat.fuse(_params(nat, kn, min_final), 1, *args)
        exp = F.fuse(*[[(int(i), float(s)) for i, s in zip(ci, cs)] for ci, cs in chans], kn)
        got = [(int(ids[0, r]), float(vals[0, r, 0])) for r in range(len(exp))]
        # equal scores may be permuted only inside exact-tie groups, and the oracle and the kernel
        # both break ties by first appearance -> identical order expected
        assert got == [(h["id"], h["score"]) for h in exp], (it, kn)
        assert count[0] == sum(1 for h in exp if h["score"] >= min_final)
        for r, h in enumerate(exp):
            sb = h["breakdown"]
            assert vals[0, r, 1] == sb["rrf_norm"] and vals[0, r, 2] == sb["weighted_sum"]
            assert [vals[0, r, 3], vals[0, r, 4], vals[0, r, 5]] == [sb["dense_norm"], sb["bm25_norm"], sb["colbert_norm"]]
            assert [vals[0, r, 6 + c] for c in range(3)] == [sb["channel_contrib"][n] for n in ("dense", "bm25", "colbert")]


def test_bm25_fuzz_vs_oracle(nat):
    from oracle import bm25 as OB
    rng = np.random.default_rng(4242)
    for it in range(12):
        n_docs = int(rng.choice([1, 2, 63, 64, 65, 1000, 1025, 4096, 4097, 10000]))
        vocab = int(rng.choice([3, 50, 800]))
        docs, words = toy_corpus(rng, n_docs, vocab, int(rng.choice([1, 12, 90])))
        ob, csr, gi = bm25_pair(nat, docs)
        k = int(rng.choice([1, 10, 100, 256]))
        queries = [[words[j] for j in rng.integers(0, vocab, size=int(rng.integers(0, 9)))] + (["?"] if q % 2 else [])
                   for q in range(5)]
        tid = [[csr["vocab"].get(t, -1) for t in q] for q in queries]
        s, i = gi.search(tid, k)
        full = gi.get_scores(tid)
        for qn, q in enumerate(queries):
            assert np.array_equal(full[qn], ob.get_scores(q))
            exp = OB.search(ob, q, k)
            kk = min(k, n_docs)
            assert i[qn, :kk].tolist() == [e[0] for e in exp] and s[qn, :kk].tolist() == [e[1] for e in exp]
        gi.close()


def test_bm25_rank_bm25_readme_example(nat):
    """The product on the one published rank_bm25 vector (see tests/test_oracle_selfcheck.py)."""
    from test_oracle_selfcheck import RANK_BM25_README_CORPUS, RANK_BM25_README_QUERY, RANK_BM25_README_SCORES
    ob, csr, gi = bm25_pair(nat, [d.split(" ") for d in RANK_BM25_README_CORPUS])
    tid = [[csr["vocab"].get(t, -1) for t in RANK_BM25_README_QUERY.split(" ")]]
    full = gi.get_scores(tid)[0]
    assert [round(float(x), 8) for x in full] == RANK_BM25_README_SCORES
    s, i = gi.search(tid, 3)
    assert i[0].tolist() == [1, 0, 2] and round(float(s[0, 0]), 8) == 0.93729472 and s[0, 1] == 0.0
    gi.close()


def test_bm25_long_queries_token_groups(nat):
    """Queries longer than the 64-entry token table (several table fills per query), with unknown
    tokens, repeats and tokens whose lists are empty inside a slab: scores stay bit-exact."""
    from oracle import bm25 as OB
    rng = np.random.default_rng(99)
    for n_docs in (300, 5000):
        docs, words = toy_corpus(rng, n_docs, 400, 30)
        ob, csr, gi = bm25_pair(nat, docs)
        queries = []
        for qlen in (64, 65, 129, 200):
            q = [words[j] if rng.random() < 0.6 else f"unk{j}" for j in rng.integers(0, 400, size=qlen)]
            q[3:6] = [q[2]] * 3  # repeats count once per occurrence
            queries.append(q)
        queries.append(["?"] * 70)  # nothing known at all
        tid = [[csr["vocab"].get(t, -1) for t in q] for q in queries]
        full = gi.get_scores(tid)
        s, i = gi.search(tid, 10)
        for qn, q in enumerate(queries):
            assert np.array_equal(full[qn], ob.get_scores(q))
            exp = OB.search(ob, q, 10)
            assert i[qn].tolist() == [e[0] for e in exp] and s[qn].tolist() == [e[1] for e in exp]
        gi.close()


def test_dense_score_rows_and_edges(nat):
    rng = np.random.default_rng(8)
    X = unit_rows(rng, 500, 768)
    Q = unit_rows(rng, 3, 768)
    idx = nat.DenseIndex(X)
    rows = rng.integers(-2, 520, size=(3, 37))
    got = idx.score_rows(Q, rows)
    ref = (Q.astype(np.float64) @ X.astype(np.float64).T)
    for q in range(3):
        for j, r in enumerate(rows[q]):
            if 0 <= r < 500:
                assert abs(got[q, j] - ref[q, r]) <= TOL
            else:
                assert got[q, j] == -np.finfo(np.float32).max
    # empty index: every slot is padding; zero queries: no-op
    e = nat.DenseIndex(dim=768)
    s, i = e.search(Q, 5)
    assert np.all(i == -1) and np.all(s == -np.finfo(np.float32).max) and e.ntotal == 0
    e.add(X[:3])
    s, i = e.search(Q[:1], 5)
    assert sorted(i[0, :3].tolist()) == [0, 1, 2] and np.all(i[0, 3:] == -1)
    s0, i0 = idx.search(np.zeros((0, 768), np.float32), 5)
    assert s0.shape == (0, 5) and i0.shape == (0, 5)
