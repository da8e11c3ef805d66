"""Shared comparison helpers for the parity tests."""
from __future__ import annotations

import math


def tie_groups(hits, key="score"):
    """Split a ranked hit list into runs of EXACTLY equal score."""
    groups, cur = [], []
    for h in hits:
        if cur and h[key] != cur[-1][key]:
            groups.append(cur)
            cur = []
        cur.append(h)
    if cur:
        groups.append(cur)
    return groups


def assert_hits_equal_mod_ties(got, exp, *, check_breakdown=True, float_tol=0.0):
    """Same hits, same scores (bit-exact unless float_tol), same order except
    inside groups of exactly tied scores (the reference's order there depends on
    PYTHONHASHSEED)."""
    assert len(got) == len(exp), (len(got), len(exp))
    gg, eg = tie_groups(got), tie_groups(exp)
    assert [len(g) for g in gg] == [len(g) for g in eg]
    pos = 0
    for g, e in zip(gg, eg):
        gm = {h["id"]: h for h in g}
        em = {h["id"]: h for h in e}
        assert set(gm) == set(em), (sorted(gm), sorted(em))
        for cid, eh in em.items():
            gh = gm[cid]
            _close(gh["score"], eh["score"], float_tol, f"{cid}.score")
            assert gh.get("source", "retriever") == eh.get("source", "retriever"), cid
            if check_breakdown:
                _cmp_breakdown(gh["breakdown"], eh["breakdown"], float_tol, cid)
        ranks_g = sorted(h["rank"] for h in g)
        ranks_e = sorted(h["rank"] for h in e)
        assert ranks_g == ranks_e == list(range(pos + 1, pos + len(g) + 1))
        pos += len(g)


def _close(a, b, tol, what):
    if tol == 0.0:
        assert a == b or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b)), \
            f"{what}: {a!r} != {b!r}"
    else:
        assert abs(a - b) <= tol * max(1.0, abs(b)), f"{what}: {a!r} vs {b!r}"


def _cmp_breakdown(g, e, tol, cid):
    if e is None:
        assert not g
        return
    assert set(g) == set(e), (cid, sorted(g), sorted(e))
    for k, ev in e.items():
        gv = g[k]
        if isinstance(ev, float):
            _close(float(gv), ev, tol, f"{cid}.{k}")
        elif isinstance(ev, dict):
            assert set(gv) == set(ev), (cid, k)
            for kk, evv in ev.items():
                if isinstance(evv, float):
                    _close(float(gv[kk]), evv, tol, f"{cid}.{k}.{kk}")
                else:
                    assert gv[kk] == evv, (cid, k, kk)
        elif k == "channel":
            # order among channels with exactly equal contribution is by name (deterministic)
            assert list(gv) == list(ev), (cid, gv, ev)
        else:
            assert gv == ev, (cid, k, gv, ev)
