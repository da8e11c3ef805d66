"""ctypes binding of libamdretrieval.so (include/amdretrieval.h).

There is NO CPU fallback: if the shared library is missing or a call fails the
error is raised to the caller.  The GIL is released for the duration of every
native call (ctypes.CDLL), so concurrent searches from the retrieval service's
thread pool overlap their host-side work.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

LIB_NAME = "libamdretrieval.so"
MAX_K = 256
MAX_DIM = 1024
MAXSIM_DIM = 128
MAXSIM_QLEN = 32
FUSE_NVALS = 9
FUSE_METHODS = {"rrf_norm_blend": 0, "rrf": 1, "wrrf": 2, "weighted_sum": 3}
FV = dict(score=0, rrf_norm=1, weighted_sum=2, dense_norm=3, bm25_norm=4, colbert_norm=5,
          contrib_dense=6, contrib_bm25=7, contrib_colbert=8)

# every symbol include/amdretrieval.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "amdr_last_error", "amdr_version", "amdr_device_count", "amdr_device_name",
    "amdr_dense_create", "amdr_dense_create_from_device", "amdr_dense_add", "amdr_dense_ntotal", "amdr_dense_dim",
    "amdr_dense_reserve", "amdr_dense_search", "amdr_dense_search_device", "amdr_dense_search_fuse_device", "amdr_hybrid_small_device", "amdr_dense_small_create", "amdr_dense_small_approx_device", "amdr_dense_small_destroy", "amdr_dense_two_pass_fallbacks", "amdr_dense_read_rows", "amdr_dense_score_rows",
    "amdr_dense_plan_info", "amdr_dense_workspace_plan", "amdr_dense_hi_counters", "amdr_dense_profile_begin", "amdr_dense_profile_end", "amdr_dense_destroy",
    "amdr_bm25_create", "amdr_bm25_ndocs", "amdr_bm25_reserve", "amdr_bm25_search", "amdr_bm25_search_device",
    "amdr_bm25_scores", "amdr_bm25_destroy",
    "amdr_tokenizer_create", "amdr_tokenizer_encode", "amdr_tokenizer_encode_joined", "amdr_tokenizer_encode_ptrs", "amdr_tokenizer_spans", "amdr_tokenizer_destroy",
    "amdr_maxsim_create", "amdr_maxsim_ndocs", "amdr_maxsim_plan_info", "amdr_maxsim_reserve", "amdr_maxsim_search",
    "amdr_maxsim_search_device", "amdr_maxsim_scores", "amdr_maxsim_destroy",
    "amdr_fuse", "amdr_fuse_device", "amdr_rerank_blend", "amdr_rerank_blend_device", "amdr_fuse_compact_device",
    "amdr_merge_topk_f32_device", "amdr_merge_topk_f64_device",
    "amdr_shard_row_words", "amdr_shard_pack_device", "amdr_shard_merge_device",
)


# argument kinds of every export, in header order: P = pointer (host or device, or an opaque handle /
# handle out-parameter / hipStream_t passed as void*), i = int32_t, l = int64_t, d = double.
# tests/test_abi.py parses include/amdretrieval.h and checks this table against the prototypes, so a
# wrapper can no longer pass a Python int where the ABI wants 64 bits (or the reverse) unnoticed.
SIGNATURES = {
    "amdr_last_error": "", "amdr_version": "", "amdr_device_count": "P", "amdr_device_name": "iPi",
    "amdr_dense_create": "PliiP", "amdr_dense_create_from_device": "PliiP", "amdr_dense_add": "PPl",
    "amdr_dense_ntotal": "PP", "amdr_dense_dim": "PP", "amdr_dense_reserve": "Pii", "amdr_dense_search": "PPiiPP",
    "amdr_dense_search_device": "PPiiPPP", "amdr_dense_search_fuse_device": "PPiiPPPPiPPPPPPPP", "amdr_hybrid_small_device": "PPPPPiiiPPPPPPPPPPPP", "amdr_dense_small_create": "PP", "amdr_dense_small_approx_device": "PPiPlPP", "amdr_dense_small_destroy": "P", "amdr_dense_two_pass_fallbacks": "PP", "amdr_dense_read_rows": "PllP", "amdr_dense_score_rows": "PPiPiP",
    "amdr_dense_plan_info": "PiiPi", "amdr_dense_workspace_plan": "liiiP", "amdr_dense_hi_counters": "PP", "amdr_dense_profile_begin": "Pi", "amdr_dense_profile_end": "PPP", "amdr_dense_destroy": "P",
    "amdr_bm25_create": "PPPPPlldddiP", "amdr_bm25_ndocs": "PP", "amdr_bm25_reserve": "Piil",
    "amdr_bm25_search": "PPPiiPP", "amdr_bm25_search_device": "PPPiiPPP", "amdr_bm25_scores": "PPPiP",
    "amdr_bm25_destroy": "P",
    "amdr_tokenizer_create": "PPlP", "amdr_tokenizer_encode": "PPPiPlPP", "amdr_tokenizer_encode_joined": "PPliPlPP", "amdr_tokenizer_encode_ptrs": "PPPiPlPP", "amdr_tokenizer_spans": "PlPPiP",
    "amdr_tokenizer_destroy": "P",
    "amdr_maxsim_create": "PPliiP", "amdr_maxsim_ndocs": "PP", "amdr_maxsim_plan_info": "PiPi", "amdr_maxsim_reserve": "Pii",
    "amdr_maxsim_search": "PPiiiPP", "amdr_maxsim_search_device": "PPiiiPPP", "amdr_maxsim_scores": "PPiiP",
    "amdr_maxsim_destroy": "P",
    "amdr_fuse": "Pi" + "PPi" * 3 + "PPPP", "amdr_fuse_device": "Pi" + "PPiP" * 3 + "PPPP" + "iP",
    "amdr_rerank_blend": "iiPPPPPidP", "amdr_rerank_blend_device": "iiPPPPPidPiP",
    "amdr_fuse_compact_device": "iiiPPPPPPPPiP",
    "amdr_merge_topk_f32_device": "PPiiiiPPiP", "amdr_merge_topk_f64_device": "PPiiiiPPiP",
    "amdr_shard_row_words": "PiP", "amdr_shard_pack_device": "PiilPiP", "amdr_shard_merge_device": "PiiPiiP",
}
_KIND = {"P": C.c_void_p, "i": C.c_int32, "l": C.c_int64, "d": C.c_double}


class NativeError(RuntimeError):
    """A libamdretrieval call returned a non-zero status."""


class FuseParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("rrf_k", C.c_int32), ("alpha", C.c_double), ("w_dense", C.c_double),
                ("w_bm25", C.c_double), ("w_colbert", C.c_double), ("min_final_score", C.c_double)]


class ShardChan(C.Structure):
    """amdr_shard_chan_t: one channel of the shard exchange (device pointers)."""
    _fields_ = [("scores", C.c_void_p), ("ids", C.c_void_p), ("k", C.c_int32), ("f64", C.c_int32)]


_lib: Optional[C.CDLL] = None


_pystr = False


def _pystrings():
    """The CPython helper next to the library (lib/_amdr_pystrings.so: UTF-8 views of a list of str without copies), or
    None when it was not built — the tokeniser then takes the joined-blob form.  Host glue only: no compute."""
    global _pystr
    if _pystr is False:
        _pystr = None
        p = lib_path().parent / "_amdr_pystrings.so"
        if p.exists() and os.environ.get("AMDR_NO_PYSTRINGS") != "1":
            try:
                import importlib.machinery
                import importlib.util
                loader = importlib.machinery.ExtensionFileLoader("_amdr_pystrings", str(p))
                spec = importlib.util.spec_from_loader("_amdr_pystrings", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _pystr = mod
            except Exception:  # noqa: BLE001 - glue is optional
                _pystr = None
    return _pystr


def lib_path() -> Path:
    env = os.environ.get("AMDR_LIB")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / LIB_NAME


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  If torch is
    # going to be used in this process it must be loaded FIRST so that this library's
    # NEEDED libamdhip64.so.7 resolves to the runtime torch already mapped: one HIP runtime
    # per process, and torch device pointers / stream handles are then valid here.
    if os.environ.get("AMDR_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 - torch is optional for the numpy-only API
            pass
    p = lib_path()
    if not p.exists():
        raise NativeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C legal-rag_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(str(p))
    lib.amdr_last_error.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)  # raises AttributeError if a declared symbol is missing
        if name != "amdr_last_error":
            fn.restype = C.c_int
        fn.argtypes = [_KIND[k] for k in SIGNATURES[name]]
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().amdr_last_error()
        raise NativeError(f"{what} failed (status {rc}): {msg.decode('utf-8', 'replace') if msg else ''}")


def device_count() -> int:
    n = C.c_int32(0)
    rc = load().amdr_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    _check(load().amdr_device_name(C.c_int32(device), buf, C.c_int32(256)), "amdr_device_name")
    return buf.value.decode()


def dense_workspace_plan(n: int, d: int, nq: int, k: int) -> Tuple[Tuple[int, int, int], Tuple[int, int, int]]:
    """(reserved, used by the largest pass) workspace bytes of one batched dense search — host-only arithmetic."""
    out = (C.c_int64 * 6)()
    _check(load().amdr_dense_workspace_plan(C.c_int64(n), C.c_int32(d), C.c_int32(nq), C.c_int32(k), out),
           "amdr_dense_workspace_plan")
    v = [int(x) for x in out]
    return tuple(v[:3]), tuple(v[3:])


def _p(a: Optional[np.ndarray], ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def _vp(ptr: int):
    return C.c_void_p(int(ptr) if ptr else 0)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


# ---------------------------------------------------------------------------
class DenseIndex:
    """Exact inner-product index resident in HBM (replaces faiss IndexFlatIP /
    IndexHNSWFlat behind `index.search`, dense_retriever.py:42)."""

    def __init__(self, X: Optional[np.ndarray] = None, *, device: int = 0, dim: Optional[int] = None,
                 device_ptr: Optional[int] = None, n: Optional[int] = None, keepalive=None):
        lib = load()
        self._h = C.c_void_p()
        self.device = int(device)
        self._keepalive = keepalive
        if device_ptr is not None:
            assert n is not None and dim is not None
            _check(lib.amdr_dense_create_from_device(_vp(device_ptr), C.c_int64(n), C.c_int32(dim),
                                                     C.c_int32(device), C.byref(self._h)),
                   "amdr_dense_create_from_device")
            self.d = int(dim)
        else:
            if X is None:
                X = np.zeros((0, int(dim)), dtype=np.float32)
            X = _c(X, np.float32)
            if X.ndim != 2:
                raise ValueError("X must be [n, d]")
            self.d = int(X.shape[1])
            _check(lib.amdr_dense_create(_p(X, C.c_float), C.c_int64(X.shape[0]), C.c_int32(self.d),
                                         C.c_int32(device), C.byref(self._h)), "amdr_dense_create")

    @property
    def ntotal(self) -> int:
        n = C.c_int64(0)
        _check(load().amdr_dense_ntotal(self._h, C.byref(n)), "amdr_dense_ntotal")
        return int(n.value)

    def add(self, X: np.ndarray) -> None:
        X = _c(X, np.float32)
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError(f"add: expected [*, {self.d}]")
        _check(load().amdr_dense_add(self._h, _p(X, C.c_float), C.c_int64(X.shape[0])), "amdr_dense_add")

    def reserve(self, nq_max: int, k_max: int) -> None:
        _check(load().amdr_dense_reserve(self._h, C.c_int32(nq_max), C.c_int32(k_max)), "amdr_dense_reserve")

    def search(self, Q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """faiss-shaped: (scores f32[nq,k], ids i64[nq,k]), -1 padded."""
        Q = _c(Q, np.float32)
        if Q.ndim == 1:
            Q = Q[None, :]
        if Q.shape[1] != self.d:
            raise ValueError(f"search: query dim {Q.shape[1]} != index dim {self.d}")
        nq = Q.shape[0]
        scores = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        _check(load().amdr_dense_search(self._h, _p(Q, C.c_float), C.c_int32(nq), C.c_int32(k),
                                        _p(scores, C.c_float), _p(ids, C.c_int64)), "amdr_dense_search")
        return scores, ids

    def search_device(self, q_ptr: int, nq: int, k: int, scores_ptr: int, ids_ptr: int, stream: int = 0) -> None:
        _check(load().amdr_dense_search_device(self._h, _vp(q_ptr), C.c_int32(nq), C.c_int32(k), _vp(scores_ptr),
                                               _vp(ids_ptr), _vp(stream)), "amdr_dense_search_device")

    def two_pass_fallbacks(self) -> int:
        """Queries of the two-pass long-batch searches so far that re-scored their whole row (amdr_dense_two_pass_fallbacks)."""
        out = C.c_int64(0)
        _check(load().amdr_dense_two_pass_fallbacks(self._h, C.byref(out)), "amdr_dense_two_pass_fallbacks")
        return int(out.value)

    def search_fuse_device(self, params: "FuseParams", q_ptr: int, nq: int, k: int, bm25, dense_row2uid: int,
                           scores_ptr: int, ids_ptr: int, out_ids: int, out_vals: int, out_mask: int, out_count: int,
                           stream: int = 0) -> None:
        """search_device + fuse_device(dense, bm25) as one native call (amdr_dense_search_fuse_device).
        bm25 = (ids_ptr, scores_ptr, kb, row2uid_ptr | 0)."""
        _check(load().amdr_dense_search_fuse_device(
            self._h, _vp(q_ptr), C.c_int32(nq), C.c_int32(k), C.byref(params), _vp(dense_row2uid), _vp(bm25[0]),
            _vp(bm25[1]), C.c_int32(bm25[2]), _vp(bm25[3]), _vp(scores_ptr), _vp(ids_ptr), _vp(out_ids), _vp(out_vals),
            _vp(out_mask), _vp(out_count), _vp(stream)), "amdr_dense_search_fuse_device")

    def score_rows(self, Q: np.ndarray, rows: np.ndarray) -> np.ndarray:
        """out[q, j] = <Q[q], X[rows[q, j]]> (rows outside [0, n) -> -FLT_MAX)."""
        Q = _c(Q, np.float32)
        if Q.ndim == 1:
            Q = Q[None, :]
        rows = _c(rows, np.int64).reshape(Q.shape[0], -1)
        out = np.empty(rows.shape, dtype=np.float32)
        _check(load().amdr_dense_score_rows(self._h, _p(Q, C.c_float), C.c_int32(Q.shape[0]), _p(rows, C.c_int64),
                                            C.c_int32(rows.shape[1]), _p(out, C.c_float)), "amdr_dense_score_rows")
        return out

    def read_rows(self, row0: int, nrows: int) -> np.ndarray:
        out = np.empty((nrows, self.d), dtype=np.float32)
        _check(load().amdr_dense_read_rows(self._h, C.c_int64(row0), C.c_int64(nrows), _p(out, C.c_float)),
               "amdr_dense_read_rows")
        return out

    def plan_info(self, nq: int, k: int) -> str:
        """Kernels a search of nq queries at depth k launches on this index, and the cut of the work."""
        buf = C.create_string_buffer(1024)
        _check(load().amdr_dense_plan_info(self._h, C.c_int32(nq), C.c_int32(k), buf, C.c_int32(512)),
               "amdr_dense_plan_info")
        return buf.value.decode()

    def hi_counters(self) -> Tuple[int, int, int, bool, int, int]:
        """(queries that took the fp16 first pass of large scans, those it could not resolve, current width level 0-2,
        pass still in use, passes, passes that also ran the exact chain).  Synchronises the device."""
        out = (C.c_int64 * 6)()
        _check(load().amdr_dense_hi_counters(self._h, out), "amdr_dense_hi_counters")
        return int(out[0]), int(out[1]), int(out[2]), bool(out[3]), int(out[4]), int(out[5])

    def profile_begin(self, max_launches: int) -> None:
        _check(load().amdr_dense_profile_begin(self._h, C.c_int32(max_launches)), "amdr_dense_profile_begin")

    def profile_end(self) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_int32(0)
        _check(load().amdr_dense_profile_end(self._h, C.byref(ms), C.byref(n)), "amdr_dense_profile_end")
        return float(ms.value), int(n.value)

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            load().amdr_dense_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
class BM25Index:
    """Okapi BM25 over term-major CSR postings (replaces BM25Okapi.get_scores +
    the Python sort, bm25_retriever.py:74-75)."""

    def __init__(self, term_ptr, post_doc, post_tf, idf, doc_len, avgdl: float, k1: float = 1.5, b: float = 0.75,
                 *, device: int = 0):
        lib = load()
        self._arrs = (_c(term_ptr, np.int64), _c(post_doc, np.int32), _c(post_tf, np.int32), _c(idf, np.float64),
                      _c(doc_len, np.int32))
        tp, pd, pt, idf_, dl = self._arrs
        self.n_terms = int(tp.shape[0] - 1)
        self.n_docs = int(dl.shape[0])
        self.device = int(device)
        self._h = C.c_void_p()
        _check(lib.amdr_bm25_create(_p(tp, C.c_int64), _p(pd, C.c_int32), _p(pt, C.c_int32), _p(idf_, C.c_double),
                                    _p(dl, C.c_int32), C.c_int64(self.n_terms), C.c_int64(self.n_docs),
                                    C.c_double(avgdl), C.c_double(k1), C.c_double(b), C.c_int32(device),
                                    C.byref(self._h)), "amdr_bm25_create")
        self._arrs = None  # the library holds its own device copy

    @staticmethod
    def pack_queries(queries: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
        q_ptr = np.zeros(len(queries) + 1, dtype=np.int64)
        for i, q in enumerate(queries):
            q_ptr[i + 1] = q_ptr[i] + len(q)
        q_terms = np.empty(max(int(q_ptr[-1]), 1), dtype=np.int32)
        for i, q in enumerate(queries):
            q_terms[q_ptr[i]:q_ptr[i + 1]] = np.asarray(q, dtype=np.int32).reshape(-1)
        return q_terms, q_ptr

    def reserve(self, nq_max: int, k_max: int, total_terms_max: int) -> None:
        _check(load().amdr_bm25_reserve(self._h, C.c_int32(nq_max), C.c_int32(k_max), C.c_int64(total_terms_max)),
               "amdr_bm25_reserve")

    def search(self, queries: Sequence[Sequence[int]], k: int) -> Tuple[np.ndarray, np.ndarray]:
        q_terms, q_ptr = self.pack_queries(queries)
        nq = len(queries)
        scores = np.empty((nq, k), dtype=np.float64)
        ids = np.empty((nq, k), dtype=np.int64)
        _check(load().amdr_bm25_search(self._h, _p(q_terms, C.c_int32), _p(q_ptr, C.c_int64), C.c_int32(nq),
                                       C.c_int32(k), _p(scores, C.c_double), _p(ids, C.c_int64)), "amdr_bm25_search")
        return scores, ids

    def search_device(self, q_terms_ptr: int, q_ptr_ptr: int, nq: int, k: int, scores_ptr: int, ids_ptr: int,
                      stream: int = 0) -> None:
        _check(load().amdr_bm25_search_device(self._h, _vp(q_terms_ptr), _vp(q_ptr_ptr), C.c_int32(nq), C.c_int32(k),
                                              _vp(scores_ptr), _vp(ids_ptr), _vp(stream)), "amdr_bm25_search_device")

    def get_scores(self, queries: Sequence[Sequence[int]]) -> np.ndarray:
        q_terms, q_ptr = self.pack_queries(queries)
        nq = len(queries)
        out = np.empty((nq, self.n_docs), dtype=np.float64)
        _check(load().amdr_bm25_scores(self._h, _p(q_terms, C.c_int32), _p(q_ptr, C.c_int64), C.c_int32(nq),
                                       _p(out, C.c_double)), "amdr_bm25_scores")
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            load().amdr_bm25_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
class Tokenizer:
    """Batched native query tokeniser + vocabulary lookup (include/amdretrieval.h, csrc/tokenize.cpp): the jieba.cut
    rule for text without Han characters, then term ids — one call per batch, GIL released."""

    def __init__(self, vocab: Sequence[str]):
        enc = [w.encode("utf-8") for w in vocab]
        blob = b"".join(enc)
        offs = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum([len(e) for e in enc], out=offs[1:])
        self._h = C.c_void_p()
        _check(load().amdr_tokenizer_create(C.c_char_p(blob), _p(offs, C.c_int64), C.c_int64(len(enc)), C.byref(self._h)),
               "amdr_tokenizer_create")

    def encode(self, texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(term_ids i32 [total], q_ptr i64 [n+1], needs_segmenter bool [n]) — the CSR BM25Index.search takes.
        Queries flagged needs_segmenter hold a Han character and got NO terms here.  The batch crosses into native code
        as ONE blob: the queries joined by NUL bytes and encoded once (two C-level operations however long the batch);
        a batch that itself contains a NUL takes the per-query offsets form."""
        n = len(texts)
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, bool)
        views = _pystrings()
        if views is not None:
            # zero-copy: the UTF-8 bytes where CPython keeps them (csrc/pystrings.c), pointers + lengths to the native call
            ptrs, lens = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
            cap = max(int(views.utf8_views(texts, ptrs.ctypes.data, lens.ctypes.data)), 1)
            terms = np.empty(cap, dtype=np.int32)
            q_ptr = np.empty(n + 1, dtype=np.int64)
            flags = np.empty(n, dtype=np.int32)
            _check(load().amdr_tokenizer_encode_ptrs(self._h, ptrs.ctypes.data, lens.ctypes.data, n, terms.ctypes.data, cap,
                                                     q_ptr.ctypes.data, flags.ctypes.data), "amdr_tokenizer_encode_ptrs")
            return terms[: int(q_ptr[-1])], q_ptr, flags.astype(bool)
        blob = "\0".join(t or "" for t in texts).encode("utf-8")
        cap = max(len(blob), 1)
        terms = np.empty(cap, dtype=np.int32)
        q_ptr = np.zeros(n + 1, dtype=np.int64)
        flags = np.zeros(n, dtype=np.int32)
        if blob.count(b"\0") == n - 1:
            _check(load().amdr_tokenizer_encode_joined(self._h, blob, len(blob), n, _p(terms, C.c_int32), cap,
                                                       _p(q_ptr, C.c_int64), _p(flags, C.c_int32)),
                   "amdr_tokenizer_encode_joined")
        else:
            enc = [t.encode("utf-8") for t in texts]
            blob = b"".join(enc)
            offs = np.zeros(n + 1, dtype=np.int64)
            np.cumsum([len(e) for e in enc], out=offs[1:])
            cap = max(len(blob), 1)
            terms = np.empty(cap, dtype=np.int32)
            _check(load().amdr_tokenizer_encode(self._h, C.c_char_p(blob), _p(offs, C.c_int64), C.c_int32(n),
                                                _p(terms, C.c_int32), C.c_int64(cap), _p(q_ptr, C.c_int64), _p(flags, C.c_int32)),
                   "amdr_tokenizer_encode")
        return terms[: int(q_ptr[-1])], q_ptr, flags.astype(bool)

    @staticmethod
    def cut(text: str) -> Optional[list]:
        """Token strings of one text by the native rule (None: Han text)."""
        b = text.encode("utf-8")
        cap = max(len(b), 1)
        st, en = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
        n = C.c_int32(0)
        _check(load().amdr_tokenizer_spans(C.c_char_p(b), C.c_int64(len(b)), _p(st, C.c_int32), _p(en, C.c_int32),
                                           C.c_int32(cap), C.byref(n)), "amdr_tokenizer_spans")
        if n.value < 0:
            return None
        return [b[int(a):int(e)].decode("utf-8") for a, e in zip(st[: n.value], en[: n.value])]

    def close(self) -> None:
        if self._h:
            load().amdr_tokenizer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class MaxSimIndex:
    """Exhaustive ColBERT late interaction over fp32 token embeddings."""

    def __init__(self, D: np.ndarray, doc_ptr: np.ndarray, *, device: int = 0):
        D = _c(D, np.float32)
        doc_ptr = _c(doc_ptr, np.int64)
        if D.ndim != 2:
            raise ValueError("D must be [tokens, dim]")
        self.dim = int(D.shape[1])
        self.n_docs = int(doc_ptr.shape[0] - 1)
        self.device = int(device)
        if int(doc_ptr[-1]) != D.shape[0]:
            raise ValueError("doc_ptr[-1] != number of token rows")
        self._h = C.c_void_p()
        _check(load().amdr_maxsim_create(_p(D, C.c_float), _p(doc_ptr, C.c_int64), C.c_int64(self.n_docs),
                                         C.c_int32(self.dim), C.c_int32(device), C.byref(self._h)),
               "amdr_maxsim_create")

    def plan_info(self, nq: int) -> str:
        """Kernels and arithmetic form a search of nq queries launches (no device work)."""
        buf = C.create_string_buffer(1024)
        _check(load().amdr_maxsim_plan_info(self._h, C.c_int32(nq), buf, C.c_int32(512)), "amdr_maxsim_plan_info")
        return buf.value.decode()

    def reserve(self, nq_max: int, k_max: int) -> None:
        _check(load().amdr_maxsim_reserve(self._h, C.c_int32(nq_max), C.c_int32(k_max)), "amdr_maxsim_reserve")

    def _q(self, Q):
        Q = _c(Q, np.float32)
        if Q.ndim == 2:
            Q = Q[None]
        if Q.ndim != 3 or Q.shape[2] != self.dim:
            raise ValueError(f"Q must be [nq, q_len, {self.dim}]")
        return Q

    def search(self, Q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        Q = self._q(Q)
        nq, q_len = Q.shape[0], Q.shape[1]
        scores = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        _check(load().amdr_maxsim_search(self._h, _p(Q, C.c_float), C.c_int32(nq), C.c_int32(q_len), C.c_int32(k),
                                         _p(scores, C.c_float), _p(ids, C.c_int64)), "amdr_maxsim_search")
        return scores, ids

    def search_device(self, q_ptr: int, nq: int, q_len: int, k: int, scores_ptr: int, ids_ptr: int,
                      stream: int = 0) -> None:
        _check(load().amdr_maxsim_search_device(self._h, _vp(q_ptr), C.c_int32(nq), C.c_int32(q_len), C.c_int32(k),
                                                _vp(scores_ptr), _vp(ids_ptr), _vp(stream)),
               "amdr_maxsim_search_device")

    def scores(self, Q: np.ndarray) -> np.ndarray:
        Q = self._q(Q)
        nq, q_len = Q.shape[0], Q.shape[1]
        out = np.empty((nq, self.n_docs), dtype=np.float32)
        _check(load().amdr_maxsim_scores(self._h, _p(Q, C.c_float), C.c_int32(nq), C.c_int32(q_len),
                                         _p(out, C.c_float)), "amdr_maxsim_scores")
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            load().amdr_maxsim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
def make_fuse_params(*, method: str = "rrf_norm_blend", rrf_k: int = 60, alpha: float = 0.5, w_dense: float = 0.6,
                     w_bm25: float = 0.4, w_colbert: float = 0.35, min_final_score: float = -math.inf) -> FuseParams:
    m = FUSE_METHODS.get(str(method).lower(), 0)  # unknown strings fall to the blend, like the reference's `else`
    return FuseParams(m, int(rrf_k), float(alpha), float(w_dense), float(w_bm25), float(w_colbert),
                      float(min_final_score))


def _chan(ids, scores, sdtype, nq):
    if ids is None or scores is None:
        return None, None, 0
    ids = _c(ids, np.int64).reshape(nq, -1)
    scores = _c(scores, sdtype).reshape(nq, -1)
    assert ids.shape == scores.shape
    if ids.shape[1] == 0:
        return None, None, 0
    return ids, scores, int(ids.shape[1])


def fuse(params: FuseParams, nq: int, dense=None, bm25=None, colbert=None):
    """Host-pointer fusion. Each channel is (ids i64[nq,k], scores[nq,k]) or None.
    Returns (ids [nq,max_out], vals [nq,max_out,9], mask [nq,max_out], count [nq])."""
    di, ds, kd = _chan(*(dense or (None, None)), np.float64, nq)
    bi, bs, kb = _chan(*(bm25 or (None, None)), np.float64, nq)
    ci, cs, kc = _chan(*(colbert or (None, None)), np.float64, nq)
    mo = kd + kb + kc
    out_ids = np.full((nq, max(mo, 1)), -1, dtype=np.int64)
    out_vals = np.zeros((nq, max(mo, 1), FUSE_NVALS), dtype=np.float64)
    out_mask = np.zeros((nq, max(mo, 1)), dtype=np.int32)
    out_count = np.zeros((nq,), dtype=np.int32)
    if mo == 0 or nq == 0:
        return out_ids[:, :0], out_vals[:, :0], out_mask[:, :0], out_count
    _check(load().amdr_fuse(C.byref(params), C.c_int32(nq), _p(di, C.c_int64), _p(ds, C.c_double), C.c_int32(kd),
                            _p(bi, C.c_int64), _p(bs, C.c_double), C.c_int32(kb), _p(ci, C.c_int64),
                            _p(cs, C.c_double), C.c_int32(kc), _p(out_ids, C.c_int64), _p(out_vals, C.c_double),
                            _p(out_mask, C.c_int32), _p(out_count, C.c_int32)), "amdr_fuse")
    return out_ids, out_vals, out_mask, out_count


def fuse_device(params: FuseParams, nq: int, dense, bm25, colbert, out_ids: int, out_vals: int, out_mask: int,
                out_count: int, *, device: int = 0, stream: int = 0) -> None:
    """Device-pointer fusion. Each channel = (ids_ptr, scores_ptr, k, row2uid_ptr|0) or None."""
    def un(c):
        return c if c is not None else (0, 0, 0, 0)
    d, b, c = un(dense), un(bm25), un(colbert)
    _check(load().amdr_fuse_device(C.byref(params), C.c_int32(nq),
                                   _vp(d[0]), _vp(d[1]), C.c_int32(d[2]), _vp(d[3]),
                                   _vp(b[0]), _vp(b[1]), C.c_int32(b[2]), _vp(b[3]),
                                   _vp(c[0]), _vp(c[1]), C.c_int32(c[2]), _vp(c[3]),
                                   _vp(out_ids), _vp(out_vals), _vp(out_mask), _vp(out_count),
                                   C.c_int32(device), _vp(stream)), "amdr_fuse_device")


def rerank_blend(count: np.ndarray, ids: np.ndarray, vals: np.ndarray, mask: np.ndarray, ce_raw: np.ndarray,
                 beta: float):
    """In-place rerank blend on host arrays (copies through the device).
    ce_raw: [nq, top_n].  Returns out_rerank [nq, max_out, 2] (raw, norm)."""
    nq, max_out = ids.shape
    ce_raw = _c(ce_raw, np.float64).reshape(nq, -1)
    top_n = int(ce_raw.shape[1])
    out = np.full((nq, max_out, 2), np.nan, dtype=np.float64)
    assert ids.flags.c_contiguous and vals.flags.c_contiguous and mask.flags.c_contiguous
    count = _c(count, np.int32)
    _check(load().amdr_rerank_blend(C.c_int32(nq), C.c_int32(max_out), _p(count, C.c_int32), _p(ids, C.c_int64),
                                    _p(vals, C.c_double), _p(mask, C.c_int32), _p(ce_raw, C.c_double),
                                    C.c_int32(top_n), C.c_double(beta), _p(out, C.c_double)), "amdr_rerank_blend")
    return out


def rerank_blend_device(nq: int, max_out: int, count: int, ids: int, vals: int, mask: int, ce_raw: int, top_n: int,
                        beta: float, out_rerank: int, *, device: int = 0, stream: int = 0) -> None:
    _check(load().amdr_rerank_blend_device(C.c_int32(nq), C.c_int32(max_out), _vp(count), _vp(ids), _vp(vals),
                                           _vp(mask), _vp(ce_raw), C.c_int32(top_n), C.c_double(beta),
                                           _vp(out_rerank), C.c_int32(device), _vp(stream)),
           "amdr_rerank_blend_device")


class DenseSmallApprox:
    """The fp16 first pass over a short corpus on its own (amdr_dense_small_*; tests and measurements — the search calls
    run it inside): approximate scores of every (query, row) with a proven per-query bound on their distance from the
    exact dot product."""

    def __init__(self, dense: "DenseIndex"):
        self._h = C.c_void_p()
        self._dense = dense  # must outlive this handle
        _check(load().amdr_dense_small_create(dense._h, C.byref(self._h)), "amdr_dense_small_create")

    def approx_device(self, q_ptr: int, nq: int, s_ptr: int, ld: int, eps_ptr: int = 0, stream: int = 0) -> None:
        _check(load().amdr_dense_small_approx_device(self._h, _vp(q_ptr), C.c_int32(nq), _vp(s_ptr), C.c_int64(ld), _vp(eps_ptr),
                                                     _vp(stream)), "amdr_dense_small_approx_device")

    def close(self) -> None:
        if self._h:
            load().amdr_dense_small_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def hybrid_small_plan(dense: "DenseIndex", bm25: "BM25Index", nq: int, kd: int, kb: int, dense_row2uid: int,
                      bm25_row2uid: int, dense_scores: int, dense_ids: int, bm25_scores: int, bm25_ids: int, out_ids: int,
                      out_vals: int, out_mask: int, out_count: int):
    """The per-shape part of an amdr_hybrid_small_device call as ready ctypes values (the serving call is issued once per
    query: building twenty ctypes objects per call costs more than the enqueue)."""
    return (load().amdr_hybrid_small_device, dense._h, bm25._h, C.c_int32(nq), C.c_int32(kd), C.c_int32(kb),
            _vp(dense_row2uid), _vp(bm25_row2uid), _vp(dense_scores), _vp(dense_ids), _vp(bm25_scores), _vp(bm25_ids),
            _vp(out_ids), _vp(out_vals), _vp(out_mask), _vp(out_count))


def hybrid_small_device(plan, params: "FuseParams", q_emb: int, q_terms: int, q_ptr: int, stream: int = 0) -> None:
    """BM25 top-k + dense top-k + fusion of 1-4 queries on a serving corpus as ONE launch (amdr_hybrid_small_device);
    every other shape runs bm25.search_device + dense.search_fuse_device inside.  Same five outputs, bit for bit."""
    fn, dh, bh, nq, kd, kb, m0, m1, ds, di, bs, bi, oi, ov, om, oc = plan
    rc = fn(dh, bh, q_emb, q_terms, q_ptr, nq, kd, kb, C.byref(params), m0, m1, ds, di, bs, bi, oi, ov, om, oc, stream)
    if rc:
        _check(rc, "amdr_hybrid_small_device")


def fuse_compact_device(nq: int, max_out: int, w: int, ids: int, vals: int, mask: int, count: int, out_rows: int,
                        out_scores: int, out_mask: int, out_count: int, *, device: int = 0, stream: int = 0) -> None:
    _check(load().amdr_fuse_compact_device(nq, max_out, w, ids, vals, mask, count, out_rows, out_scores, out_mask, out_count,
                                           device, stream), "amdr_fuse_compact_device")


def merge_topk_device(scores: int, ids: int, n_parts: int, nq: int, k_in: int, k_out: int, out_scores: int,
                      out_ids: int, *, f64: bool, device: int = 0, stream: int = 0) -> None:
    fn = load().amdr_merge_topk_f64_device if f64 else load().amdr_merge_topk_f32_device
    _check(fn(_vp(scores), _vp(ids), C.c_int32(n_parts), C.c_int32(nq), C.c_int32(k_in), C.c_int32(k_out),
              _vp(out_scores), _vp(out_ids), C.c_int32(device), _vp(stream)), "amdr_merge_topk_device")


# ---------------------------------------------------------------------------
def shard_chans(chans):
    """[(scores_ptr, ids_ptr, k, is_f64)] -> (amdr_shard_chan_t array, n): the argument block of the shard calls."""
    arr = (ShardChan * len(chans))()
    for c, (sp, ip, k, f64) in zip(arr, chans):
        c.scores, c.ids, c.k, c.f64 = int(sp) if sp else None, int(ip) if ip else None, int(k), 1 if f64 else 0
    return arr, len(chans)


def shard_row_words(ks: Sequence[int]) -> int:
    """int64 words per query row of the packed exchange buffer: sum of 2 * k_c."""
    arr, n = shard_chans([(0, 0, k, False) for k in ks])
    w = C.c_int64(0)
    _check(load().amdr_shard_row_words(arr, C.c_int32(n), C.byref(w)), "amdr_shard_row_words")
    return int(w.value)


def shard_pack_args(args, nq: int, id_offset: int, send_ptr: int, *, device: int = 0, stream: int = 0) -> None:
    _check(load().amdr_shard_pack_device(args[0], args[1], nq, id_offset, send_ptr, device, stream),
           "amdr_shard_pack_device")


def shard_merge_args(gathered_ptr: int, world: int, nq: int, args, *, device: int = 0, stream: int = 0) -> None:
    _check(load().amdr_shard_merge_device(gathered_ptr, world, nq, args[0], args[1], device, stream),
           "amdr_shard_merge_device")


def shard_pack_device(chans, nq: int, id_offset: int, send_ptr: int, *, device: int = 0, stream: int = 0) -> None:
    """chans: [(scores_ptr, local_ids_ptr, k, is_f64)] of this rank -> send [nq, row] (ONE launch, all channels)."""
    shard_pack_args(shard_chans(chans), nq, id_offset, send_ptr, device=device, stream=stream)


def shard_merge_device(gathered_ptr: int, world: int, nq: int, out_chans, *, device: int = 0, stream: int = 0) -> None:
    """gathered [world, nq, row] -> out_chans [(out_scores_ptr, out_ids_ptr, k, is_f64)] (ONE launch, all channels)."""
    shard_merge_args(gathered_ptr, world, nq, shard_chans(out_chans), device=device, stream=stream)
