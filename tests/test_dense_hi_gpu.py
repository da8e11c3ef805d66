"""The fp16 first pass of the large-scan two-level top-k (csrc/dense_hi.hip, dense.hip run_search_two_level).

AMDR_DENSE_HI=1 pins it on small matrices (with AMDR_DENSE_TWO_LEVEL unset or 1); it must return the ids and the
score BITS of the exact two-level form (AMDR_DENSE_HI=0, AMDR_DENSE_TWO_LEVEL=1) and of the full score matrix
(AMDR_DENSE_TWO_LEVEL=0): the first pass only picks candidate tiles, the scores come from the exact fp32 kernel.
Cases: every supported width, ragged last tile, 5..130 queries (chunks of 64 + remainder), k = 1..80, matrices whose
cut the rounding bound cannot separate (repeated rows, near-duplicates: the exact chain behind the device flag must
take over), extreme scales of matrix and queries, NaN rows, add() after creation."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from legal_rag_amd import _native
    _native.load()
    return _native


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def three_forms(nat, monkeypatch, X, Q, k, build=None):
    """(hi, exact two-level, full) results + the hi handle's counters"""
    out = {}
    for name, hi, tl in (("hi", "1", "1"), ("exact", "0", "1"), ("full", "0", "0")):
        monkeypatch.setenv("AMDR_DENSE_HI", hi)
        monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", tl)
        idx = build() if build else nat.DenseIndex(X)
        if name == "hi":
            assert "dense_hi_tilemax_kernel" in idx.plan_info(len(Q), k), idx.plan_info(len(Q), k)
        else:
            assert "dense_hi" not in idx.plan_info(len(Q), k)
        out[name] = idx.search(Q, k)
        if name == "hi":
            out["counters"] = idx.hi_counters()
        idx.close()
    return out


def assert_same(out, what):
    for other in ("exact", "full"):
        assert np.array_equal(out["hi"][1], out[other][1]), (what, other)
        assert np.array_equal(out["hi"][0].view(np.uint32), out[other][0].view(np.uint32)), (what, other)


def test_hi_first_pass_same_ids_and_bits(nat, monkeypatch):
    rng = np.random.default_rng(2024)
    cases = [(9017, 128, 64, 10), (9000, 256, 5, 1), (7777, 384, 33, 10), (9017, 768, 70, 10), (8200, 896, 130, 3),
             (12017, 512, 64, 80), (30011, 128, 129, 40), (7100, 640, 17, 10), (9017, 1024, 48, 10), (8200, 1024, 130, 5)]
    unresolved = 0
    for n, d, nq, k in cases:
        X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
        out = three_forms(nat, monkeypatch, X, Q, k)
        assert_same(out, (n, d, nq, k))
        took, bad = out["counters"][:2]
        assert took == nq
        unresolved += bad
    # random unit rows: the bound separates (almost) every cut — the fast path is what ran
    assert unresolved <= 2, unresolved


def test_hi_first_pass_unsupported_shapes_take_the_exact_form(nat, monkeypatch):
    rng = np.random.default_rng(5)
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    for n, d, nq, k in ((9000, 192, 33, 10), (9000, 64, 33, 10), (9000, 768, 4, 10), (9000, 768, 33, 128),
                        (1200, 768, 33, 10)):
        idx = nat.DenseIndex(unit_rows(rng, n, d))
        assert "dense_hi" not in idx.plan_info(nq, k), (n, d, nq, k)
        idx.close()


def test_hi_first_pass_unresolvable_cuts_fall_back_on_the_device(nat, monkeypatch):
    """Repeated rows / near-duplicates: more than k + 22 tiles lie within the rounding bound of the cut, the check
    raises the flag and the gated exact chain rewrites the batch — same ids (lower id first among ties), same bits."""
    rng = np.random.default_rng(77)
    base = unit_rows(rng, 300, 256)
    X = np.concatenate([base] * 40, axis=0)  # every row 40 times, 300 apart: 40 tiles tie at every rank
    Q = unit_rows(rng, 70, 256)
    out = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(out, "repeated rows")
    took, bad = out["counters"][:2]
    assert took == 70 and bad == 70
    s, i = out["hi"]
    ref = (base.astype(np.float64) @ Q.astype(np.float64).T).T
    for b in range(70):
        top = int(np.argmax(ref[b]))
        assert i[b].tolist() == [top + 300 * j for j in range(10)]
    # near-duplicates: 2 000 rows within 1e-4 of one direction, the rest random
    v = unit_rows(rng, 1, 384)
    near = v + 1e-4 * rng.standard_normal((2000, 384)).astype(np.float32)
    X = unit_rows(rng, 12000, 384)
    pos = rng.choice(12000, size=2000, replace=False)
    X[pos] = near
    Q = np.concatenate([v + 0.05 * unit_rows(rng, 20, 384), unit_rows(rng, 20, 384)])
    out = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(out, "near duplicates")
    assert out["counters"][1] >= 20  # the 20 queries aimed at the cluster cannot be resolved by the fp16 pass
    assert set(out["hi"][1][:20].ravel().tolist()) <= set(pos.tolist())


def test_hi_first_pass_scales(nat, monkeypatch):
    """Matrix and queries far from unit scale (power-of-two scaling inside the pass, the bound scales with both)."""
    rng = np.random.default_rng(8)
    for xs, qs in ((1e-18, 1.0), (3e9, 1e-7), (1.0, 5e12), (7e-4, 2e-30)):
        X = (unit_rows(rng, 9017, 256) * np.float32(xs)).astype(np.float32)
        Q = (unit_rows(rng, 40, 256) * np.float32(qs)).astype(np.float32)
        Q[3] *= np.float32(1e-6)  # per-query scale
        Q[7] = 0                  # all scores equal: lowest ids win, through the fallback
        out = three_forms(nat, monkeypatch, X, Q, 10)
        assert_same(out, (xs, qs))
        assert out["hi"][1][7].tolist() == list(range(10))
    # rows of very different norms: the largest norm drives the bound
    X = unit_rows(rng, 9017, 128) * rng.uniform(0.01, 30.0, size=(9017, 1)).astype(np.float32)
    out = three_forms(nat, monkeypatch, X.astype(np.float32), unit_rows(rng, 64, 128), 10)
    assert_same(out, "mixed norms")


def test_hi_first_pass_nan_rows_and_nan_queries(nat, monkeypatch):
    rng = np.random.default_rng(19)
    X, Q = unit_rows(rng, 9017, 128), unit_rows(rng, 33, 128)
    X[::7] = np.nan
    out = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(out, "nan rows")
    assert not np.isnan(out["hi"][0]).any() and not (out["hi"][1] % 7 == 0).any()
    Xc = unit_rows(rng, 9017, 128)
    Q[5, 3] = np.nan
    Q[6, 0] = np.inf
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    idx = nat.DenseIndex(Xc)
    hi = idx.search(Q, 10)
    idx.close()
    monkeypatch.setenv("AMDR_DENSE_HI", "0")
    idx = nat.DenseIndex(Xc)
    ex = idx.search(Q, 10)
    idx.close()
    assert np.array_equal(hi[1], ex[1])  # the exact two-level form decides what a NaN / infinite query returns
    assert np.array_equal(hi[0].view(np.uint32), ex[0].view(np.uint32))
    # an infinite component in the matrix: the statistics are not finite, the pass is not taken at all
    Xi = Xc.copy()
    Xi[100, 5] = np.inf
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    idx = nat.DenseIndex(Xi)
    assert "dense_hi" not in idx.plan_info(33, 10)
    idx.close()


def test_hi_first_pass_after_add(nat, monkeypatch):
    """add() folds the new rows into the statistics (a larger component changes the scale, a larger norm the bound)."""
    rng = np.random.default_rng(23)
    A, B = unit_rows(rng, 6000, 256), unit_rows(rng, 3017, 256) * np.float32(40.0)
    Q = unit_rows(rng, 64, 256)

    def build():
        idx = nat.DenseIndex(A)
        idx.add(B)
        return idx
    out = three_forms(nat, monkeypatch, None, Q, 10, build=build)
    assert_same(out, "add")
    assert (out["hi"][1] >= 6000).all()  # the long rows win
    one = three_forms(nat, monkeypatch, np.concatenate([A, B]), Q, 10)
    assert np.array_equal(one["hi"][1], out["hi"][1]) and np.array_equal(one["hi"][0], out["hi"][0])


def test_hi_first_pass_device_entry_and_reserve(nat, monkeypatch):
    """The `_device` entry (caller's stream, no host round trip inside) after reserve(): no allocation during the
    search, same results as the host entry."""
    import torch
    rng = np.random.default_rng(29)
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    X, Q = unit_rows(rng, 9017, 384), unit_rows(rng, 100, 384)
    idx = nat.DenseIndex(X)
    ref = idx.search(Q, 10)
    idx.reserve(100, 10)
    Qd = torch.from_numpy(Q).cuda()
    s = torch.empty(100, 10, dtype=torch.float32, device="cuda")
    i = torch.empty(100, 10, dtype=torch.int64, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    for nq in (100, 64, 37, 5):
        idx.search_device(Qd[:nq].data_ptr(), nq, 10, s.data_ptr(), i.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(i[:nq].cpu().numpy(), ref[1][:nq])
        assert np.array_equal(s[:nq].cpu().numpy(), ref[0][:nq])
    assert torch.cuda.mem_get_info()[0] == free0
    idx.close()


def _search_in_calls(idx, Q, k, per_call):
    outs = [idx.search(Q[i:i + per_call], k) for i in range(0, len(Q), per_call)]
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


def test_hi_first_pass_widens_its_cut_when_queries_stay_unresolved(nat, monkeypatch):
    """Every row 40 times: 40 tiles tie at every rank, more than the 33 candidates of level 0 but fewer than the 65 of
    level 1.  After 256 queries with > 10 % unresolved the handle moves to level 1 and the fp16 pass resolves the rest;
    results equal the exact form's before, across and after the move."""
    rng = np.random.default_rng(31)
    base = unit_rows(rng, 300, 128)
    X = np.concatenate([base] * 40, axis=0)
    Q = unit_rows(rng, 4 * 64 + 5 * 64, 128)
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    monkeypatch.setenv("AMDR_DENSE_HI", "0")
    ex_idx = nat.DenseIndex(X)
    ex = _search_in_calls(ex_idx, Q, 10, 64)
    ex_idx.close()
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    idx = nat.DenseIndex(X)
    got = _search_in_calls(idx, Q[:256], 10, 64)   # 4 calls at level 0: all unresolved
    assert idx.hi_counters() == (256, 256, 0, True, 4, 4)
    got2 = _search_in_calls(idx, Q[256:], 10, 64)  # the 5th call sees 256 of 256 unresolved: level 1 from here on
    took, bad, level, in_use, passes, flagged = idx.hi_counters()
    # (a query whose two best base rows score within 2 eps of each other has 80 tiles at its cut: still unresolved at
    # level 1 — one such query flags its whole pass, so the handle may move on to level 2 within these calls)
    assert took == 576 and 256 <= bad <= 256 + 64 and level >= 1 and in_use and passes == 9 and 4 <= flagged <= 9
    assert f"width level {level}" in idx.plan_info(64, 10)
    idx.close()
    for a, b in ((got, (ex[0][:256], ex[1][:256])), (got2, (ex[0][256:], ex[1][256:]))):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    # a pinned level neither moves nor is moved
    monkeypatch.setenv("AMDR_DENSE_HI_LEVEL", "2")
    idx = nat.DenseIndex(X)
    got3 = _search_in_calls(idx, Q, 10, 64)
    took, bad, level, in_use = idx.hi_counters()[:4]
    assert took == len(Q) and bad <= 8 and level == 2 and in_use  # three base rows within 2 eps: 120 tiles, rare
    idx.close()
    assert np.array_equal(got3[1], ex[1]) and np.array_equal(got3[0].view(np.uint32), ex[0].view(np.uint32))


def test_hi_first_pass_is_given_up_on_a_matrix_it_cannot_resolve(nat, monkeypatch):
    """Every row 150 times (more ties than the widest cut holds), the pass chosen by the library itself (no pin): levels
    0 -> 1 -> 2 -> given up, 256 queries each; the handle then runs exact passes in the same shapes.  Same results all
    the way, and a `_device` caller that reserved once never sees an allocation."""
    import torch
    rng = np.random.default_rng(37)
    base = unit_rows(rng, 1600, 128)
    X = np.concatenate([base] * 150, axis=0)  # 240 000 rows = 7 500 tiles >= 64 x 107
    Q = unit_rows(rng, 5 * 256, 128)
    monkeypatch.setenv("AMDR_DENSE_NT", "1")  # "far larger than the caches", at test size
    monkeypatch.delenv("AMDR_DENSE_HI", raising=False)
    monkeypatch.delenv("AMDR_DENSE_TWO_LEVEL", raising=False)
    idx = nat.DenseIndex(X)
    assert "dense_hi_tilemax_kernel" in idx.plan_info(256, 10)
    idx.reserve(256, 10)
    Qd = torch.from_numpy(Q).cuda()
    s = torch.empty(256, 10, dtype=torch.float32, device="cuda")
    i = torch.empty(256, 10, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    free0 = torch.cuda.mem_get_info()[0]
    got, states = [], []
    for c in range(5):
        idx.search_device(Qd[256 * c:].data_ptr(), 256, 10, s.data_ptr(), i.data_ptr(), st)
        torch.cuda.synchronize()
        got.append((s.cpu().numpy().copy(), i.cpu().numpy().copy()))
        states.append(idx.hi_counters())
    assert torch.cuda.mem_get_info()[0] == free0
    assert [x[2] for x in states] == [0, 1, 2, 2, 2] and [x[3] for x in states] == [True, True, True, False, False]
    assert [x[0] for x in states] == [256, 512, 768, 768, 768] and states[-1][1] == 768
    assert "given up" in idx.plan_info(256, 10)
    idx.close()
    monkeypatch.setenv("AMDR_DENSE_HI", "0")
    ex_idx = nat.DenseIndex(X)
    ex = ex_idx.search(Q, 10)
    ex_idx.close()
    gs, gi = np.concatenate([g[0] for g in got]), np.concatenate([g[1] for g in got])
    assert np.array_equal(gi, ex[1]) and np.array_equal(gs.view(np.uint32), ex[0].view(np.uint32))
    for b in range(0, len(Q), 97):
        assert (gi[b] % 1600 == gi[b, 0] % 1600).all() and gi[b].tolist() == sorted(gi[b].tolist())


def test_hi_first_pass_candidate_list_flushes_and_overflow(nat, monkeypatch):
    """The emitting scan stages its candidates in a wave-private LDS buffer and appends them to one flat list.  A 64-entry
    buffer (test hook) flushes after every emitting tile: the entries must arrive intact — same results, nothing
    unresolved.  A list too short for the candidates (test hook; in production: a threshold that lets too much pass)
    raises the flag: every query through the exact chain, same results."""
    rng = np.random.default_rng(41)
    X, Q = unit_rows(rng, 20011, 256), unit_rows(rng, 64, 256)
    ref = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(ref, "reference")
    monkeypatch.setenv("AMDR_DENSE_HI_WBUF", "64")
    out = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(out, "64-entry staging buffer")
    assert out["counters"][:2] == (64, 0) and out["counters"][4:] == (1, 0) and np.array_equal(out["hi"][1], ref["hi"][1])
    monkeypatch.delenv("AMDR_DENSE_HI_WBUF")
    monkeypatch.setenv("AMDR_DENSE_HI_CAP", "500")  # 64 queries x 33 candidates do not fit
    out = three_forms(nat, monkeypatch, X, Q, 10)
    assert_same(out, "overflowing candidate list")
    assert out["counters"][:2] == (64, 64) and out["counters"][4:] == (1, 1)
    monkeypatch.delenv("AMDR_DENSE_HI_CAP")
    # a constant matrix: every tile maximum ties -> every tile of every query is emitted (several flushes per wave, the
    # list's real capacity exceeded) -> exact chain -> the ten lowest ids
    Xc = np.tile(unit_rows(rng, 1, 128), (300_000, 1))
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    idx = nat.DenseIndex(Xc)
    s, i = idx.search(unit_rows(rng, 64, 128), 10)
    took, bad = idx.hi_counters()[:2]
    idx.close()
    assert (took, bad) == (64, 64)
    assert (i == np.arange(10)[None, :]).all() and (s == s[:, :1]).all()


def test_hi_first_pass_seeded_sweep_of_shapes(nat, monkeypatch):
    """Seeded sweep: n, d (every supported width), batch size (1-4 passes of 48 / 64 queries + remainders), k to 100,
    plain and scaled rows — ids and score bits of the exact two-level form every time."""
    rng = np.random.default_rng(4242)
    unresolved = took_total = 0
    for trial in range(24):
        d = int(rng.choice([128, 256, 384, 512, 640, 768, 896, 1024]))
        k = int(rng.choice([1, 2, 5, 10, 17, 40, 100]))
        kc_max = k + max(k, 96) + 1
        n = int(rng.integers(64 * kc_max + 1, 64 * kc_max + 40000))
        nq = int(rng.integers(5, 200))
        X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
        if trial % 3 == 1:
            X *= rng.uniform(0.2, 5.0, size=(n, 1)).astype(np.float32)
            Q *= np.float32(rng.uniform(1e-3, 1e3))
        out = {}
        for name, hi in (("hi", "1"), ("exact", "0")):
            monkeypatch.setenv("AMDR_DENSE_HI", hi)
            monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
            idx = nat.DenseIndex(X)
            out[name] = idx.search(Q, k)
            if name == "hi":
                assert "dense_hi_tilemax_kernel" in idx.plan_info(nq, k), (n, d, nq, k)
                c = idx.hi_counters()
                took_total += c[0]
                unresolved += c[1]
            idx.close()
        assert np.array_equal(out["hi"][1], out["exact"][1]), (trial, n, d, nq, k)
        assert np.array_equal(out["hi"][0].view(np.uint32), out["exact"][0].view(np.uint32)), (trial, n, d, nq, k)
    assert took_total > 1500 and unresolved * 20 < took_total  # the fast path is what ran


def test_hi_first_pass_against_the_cpu_oracle_directly(nat, monkeypatch):
    """The fp16 first pass + round-4 tail held to oracle/dense.py itself (exact FlatIP in numpy), not only to the other
    HIP forms: the assertions of test_kernels_gpu.check_dense — reported score == exact score of the reported id within
    1e-4, descending, the oracle's hits clearly above the cut all present, ranks equal wherever the oracle's neighbouring
    scores are separated by more than the tolerance — on pinned-hi shapes: d = 768 with 70 queries (a 64-query tile +
    remainder), d = 1 024 with 48 queries per scan, a matrix scaled by 37.5 with queries scaled by 1/1024, k = 10 and 40,
    and both tails (AMDR_DENSE_HI_TAIL=1: round 4, =0: round 3)."""
    from test_kernels_gpu import check_dense
    rng = np.random.default_rng(99)
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    cases = [(20011, 768, 70, 10, 1.0, 1.0), (16400, 1024, 48, 10, 1.0, 1.0), (18000, 1024, 100, 40, 1.0, 1.0),
             (20011, 256, 130, 10, 37.5, 1.0 / 1024.0), (20011, 384, 64, 40, 1.0, 1.0)]
    for tail in ("1", "0"):
        monkeypatch.setenv("AMDR_DENSE_HI_TAIL", tail)
        for n, d, nq, k, xs, qs in cases:
            X = (unit_rows(rng, n, d) * np.float32(xs)).astype(np.float32)
            Q = (unit_rows(rng, nq, d) * np.float32(qs)).astype(np.float32)
            probe = nat.DenseIndex(X)
            assert "dense_hi_tilemax_kernel" in probe.plan_info(nq, k), (tail, n, d, nq, k)
            probe.close()
            # check_dense's tolerance is absolute (1e-4 on unit-norm scores): bring the scaled case back to unit scale
            if xs != 1.0 or qs != 1.0:
                s, i = nat.DenseIndex(X).search(Q, k)
                from oracle import dense as OD
                es, ei = OD.flatip_topk(X, Q, k)
                assert np.array_equal(i, ei), (tail, n, d)
                assert np.max(np.abs(s - es)) <= 1e-4 * xs * qs * 4
            else:
                check_dense(nat, X, Q, k)


def test_hi_first_pass_multi_tile_passes_keep_their_thresholds(nat, monkeypatch):
    """A pass of the round-4 tail holds up to four query tiles: ONE sample launch (a block row per tile), one threshold
    per query, ONE scan launch that walks the tiles (AMDR_DENSE_HI_SCANS=split: a launch per tile).  With per-query lists
    too short to hold every tile (test hook AMDR_DENSE_HI_CAP: 1 500 entries per query on 6 250 tiles) a threshold that is
    missing or wrong for a later tile floods its lists and sends those queries through the exact chain — results stay
    right, so only the counters show it: nothing may be unresolved, for 1 - 4 tiles and a ragged last tile."""
    rng = np.random.default_rng(3)
    X, Q = unit_rows(rng, 200_000, 256), unit_rows(rng, 256, 256)
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    monkeypatch.setenv("AMDR_DENSE_HI", "0")
    idx = nat.DenseIndex(X)
    ref = idx.search(Q, 10)
    idx.close()
    monkeypatch.setenv("AMDR_DENSE_HI", "1")
    monkeypatch.setenv("AMDR_DENSE_HI_CAP", str(64 * 1500))
    for scans in ("one", "split"):
        monkeypatch.setenv("AMDR_DENSE_HI_SCANS", scans)
        for m in (64, 100, 128, 192, 256):
            idx = nat.DenseIndex(X)
            s, i = idx.search(Q[:m], 10)
            took, bad, _, in_use, passes, flagged = idx.hi_counters()
            idx.close()
            assert (took, bad, flagged, in_use) == (m, 0, 0, True) and passes == (m + 63) // 64, (scans, m, took, bad, passes, flagged)
            assert np.array_equal(i, ref[1][:m]) and np.array_equal(s.view(np.uint32), ref[0][:m].view(np.uint32)), (scans, m)
