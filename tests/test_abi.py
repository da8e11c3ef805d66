"""The C-ABI shared library loads and exports every symbol include/amdretrieval.h
declares (no compute calls: this runs without a GPU)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    src = (ROOT / "include" / "amdretrieval.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amdr_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_in_tree():
    from legal_rag_amd import _native
    p = _native.lib_path()
    assert p.exists(), f"{p} missing: run __graft_entry__.build()"
    assert ROOT in p.parents


def test_every_declared_symbol_is_exported():
    from legal_rag_amd import _native
    lib = ctypes.CDLL(str(_native.lib_path()))
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in amdretrieval.h but not exported"


def test_binding_table_matches_header():
    from legal_rag_amd import _native
    assert sorted(_native.EXPORTS) == header_symbols()


def header_prototypes():
    """name -> argument kinds ('P' pointer, 'i' int32_t, 'l' int64_t, 'd' double) parsed from the header."""
    src = (ROOT / "include" / "amdretrieval.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(amdr_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        kinds = ""
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                kinds += "P"
            elif re.search(r"\bint64_t\b", a):
                kinds += "l"
            elif re.search(r"\bint32_t\b|\bint\b", a):
                kinds += "i"
            elif re.search(r"\bdouble\b", a):
                kinds += "d"
            else:
                raise AssertionError(f"{name}: unrecognised parameter '{a}'")
        out[name] = kinds
    return out


def test_argtypes_table_matches_header_prototypes():
    """Every export has ctypes argtypes, and they are the header's parameter kinds (no call depends
    on a wrapper remembering to wrap a Python int in the right width)."""
    from legal_rag_amd import _native
    protos = header_prototypes()
    assert sorted(protos) == sorted(_native.EXPORTS)
    for name, kinds in protos.items():
        assert _native.SIGNATURES[name] == kinds, (name, _native.SIGNATURES[name], kinds)
    lib = _native.load()
    for name in _native.EXPORTS:
        assert getattr(lib, name).argtypes is not None and len(getattr(lib, name).argtypes) == len(protos[name])


def test_version_and_error_string_without_gpu():
    from legal_rag_amd import _native
    lib = _native.load()
    assert lib.amdr_version() >= 100
    # a bad-argument call fails loudly with a message and never touches a device
    rc = lib.amdr_dense_create(None, ctypes.c_int64(4), ctypes.c_int32(770), ctypes.c_int32(0), None)
    assert rc == -1
    assert b"out is null" in lib.amdr_last_error()


def test_no_cpu_fallback_in_product():
    """The product package must not import the oracle."""
    pkg = ROOT / "legal-rag_amd"
    for py in pkg.rglob("*.py"):
        txt = py.read_text(encoding="utf-8")
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{py} imports the oracle"


def test_dense_workspace_covers_every_pass(monkeypatch):
    """A batched dense search reserves its workspace once and then runs passes of the full chunk and a remainder:
    the slab-list space slabs(m) * m * k * 8 is NOT monotone in the pass size m (13 M rows, k = 10: 96 queries take
    22 slabs = 168 960 B, the 89-query remainder 24 slabs = 170 880 B), so the reservation must be the maximum over
    the passes.  Host-only arithmetic (amdr_dense_workspace_plan): no device."""
    from legal_rag_amd import _native
    checked = two_level = 0
    for n in (13_000_000, 12_345_678, 20_000_000, 40_000_000, 3_000_000, 600_000):
        for k in (1, 10, 80, 256):
            for nq in (5, 37, 95, 96, 97, 100, 131, 185, 191, 192, 193, 250, 1000):
                res, used = _native.dense_workspace_plan(n, 768, nq, k)
                assert all(u <= r for u, r in zip(used, res)), (n, k, nq, res, used)
                checked += 1
                two_level += int(res[2] > 0)
    assert checked > 300 and two_level > 50
    # the advisor's worked example, pinned: remainder 89 needs more list space than the chunk of 96
    monkeypatch.setenv("AMDR_DENSE_TWO_LEVEL", "1")
    monkeypatch.setenv("AMDR_DENSE_HI", "0")  # the exact first pass (32-query tiles, chunks of 96)
    res, used = _native.dense_workspace_plan(13_000_000, 768, 185, 10)
    assert used[1] == 170880 and res[1] >= used[1]
    # the fp16 first pass (chunks of 64, k + 23 candidate tiles per query) and the exact chain behind its flag
    for hi in ("0", "1"):
        monkeypatch.setenv("AMDR_DENSE_HI", hi)
        for n in (13_000_000, 600_000, 40_000):
            for d in (128, 384, 768, 896, 1024):
                for k in (1, 10, 80, 127, 128):
                    for nq in (5, 37, 64, 65, 100, 129):
                        res, used = _native.dense_workspace_plan(n, d, nq, k)
                        assert all(u <= r for u, r in zip(used, res)), (hi, n, d, k, nq, res, used)
