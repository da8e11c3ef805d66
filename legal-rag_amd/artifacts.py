"""Readers / writers for the index artifacts under data/index/<lang>/…

The artifact layout stays identical to the reference's (SURVEY.md §8 a-12,
legalrag/config.py:197-204):
    faiss/faiss.index        FAISS binary index container
    faiss/faiss_meta.jsonl   one LawChunk JSON per line (row i <-> chunk i)
    bm25.pkl                 pickle {"bm25": BM25Okapi, "chunks": [dict, …]}
    colbert/colbert_meta.jsonl            {"pid", "chunk"} per line
    colbert/<experiment>/indexes/<name>/  ColBERT index directory

faiss.index  — written as a FAISS `IndexFlatIP` ("IxFI") container, which the
reference's `faiss.read_index` loads and searches unchanged (it only sets
`hnsw.efSearch` when the attribute exists, vector_store.py:115-117).  Read:
flat ("IxFI"/"IxF2"/"IxFl") and HNSW-flat ("IHNf"/"IHN2", what the reference's
builder writes, faiss_builder.py:84-96) containers; from an HNSW container only
the flat fp32 storage is used — the graph is what the exact scan replaces.
FAISS's io format is restated from its published source (faiss/impl/index_write.cpp,
faiss >= 1.7.4) [from memory — faiss is absent here; verify against a real file].

colbert/<experiment>/indexes/<name>/ — this build writes its own exact fp32 token store
(`amdr_tokens.npz`); a directory written by colbert-ai's Indexer (PLAID: centroids + 4-bit residual
codes, colbert_builder.py:120-134) is READ and decompressed once into the same token store
(`read_plaid_index`), so a reference-built ColBERT index is served unchanged.  colbert-ai's on-disk
layout and residual codec are restated from its published source (colbert/indexing/codecs/
residual.py, collection_indexer.py; colbert-ai >= 0.2.19) [from memory — colbert-ai is absent
here; verify against a real index].

bm25.pkl — written so that it unpickles as `rank_bm25.BM25Okapi` under the
reference and read with a RESTRICTED unpickler (a pickle is code: only the
handful of globals such a file legitimately contains are resolvable).
"""
from __future__ import annotations

import io
import json
import pickle
import struct
import sys
import types
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .bm25_model import BM25Okapi
from .schemas import LawChunk

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _fourcc(s: str) -> int:
    b = s.encode("ascii")
    return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)


_FLAT = {_fourcc("IxFI"): METRIC_INNER_PRODUCT, _fourcc("IxF2"): METRIC_L2, _fourcc("IxFl"): None}
_HNSW_FLAT = {_fourcc("IHNf"), _fourcc("IHN2")}


# ---------------------------------------------------------------------------
# faiss.index
# ---------------------------------------------------------------------------
def write_faiss_flat_ip(path: Path, X: np.ndarray) -> None:
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, d = X.shape
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(path.suffix + ".tmp")
    with tmp.open("wb") as f:
        f.write(struct.pack("<I", _fourcc("IxFI")))
        f.write(struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, METRIC_INNER_PRODUCT))
        f.write(struct.pack("<Q", n * d))
        f.write(X.tobytes())
    tmp.replace(path)


class _Reader:
    def __init__(self, data: bytes):
        self.b = memoryview(data)
        self.o = 0

    def take(self, fmt: str):
        sz = struct.calcsize(fmt)
        if self.o + sz > len(self.b):
            raise ValueError("faiss index: truncated file")
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += sz
        return v

    def vector(self, itemsize: int) -> memoryview:
        (n,) = self.take("<Q")
        nbytes = n * itemsize
        if nbytes > len(self.b) - self.o:
            raise ValueError("faiss index: vector length exceeds file size")
        v = self.b[self.o:self.o + nbytes]
        self.o += nbytes
        return v


def _read_header(r: _Reader) -> Tuple[int, int, int]:
    d, ntotal, _d1, _d2, _trained, metric = r.take("<iqqq?i")
    if metric > 1:
        r.take("<f")  # metric_arg
    if d <= 0 or ntotal < 0:
        raise ValueError("faiss index: bad header")
    return d, ntotal, metric


def _read_flat(r: _Reader) -> Tuple[np.ndarray, int]:
    (h,) = r.take("<I")
    if h not in _FLAT:
        raise ValueError(f"faiss index: storage fourcc 0x{h:08x} is not a flat fp32 index")
    d, ntotal, metric = _read_header(r)
    raw = r.vector(4)
    X = np.frombuffer(raw, dtype=np.float32)
    if X.size != ntotal * d:
        raise ValueError("faiss index: flat storage size mismatch")
    return X.reshape(ntotal, d).copy(), metric


def read_faiss_index(path: Path) -> Tuple[np.ndarray, int]:
    """-> (X fp32 [ntotal, d], metric).  Raises ValueError on anything else."""
    data = Path(path).read_bytes()
    r = _Reader(data)
    (h,) = r.take("<I")
    if h in _FLAT:
        r.o = 0
        return _read_flat(r)
    if h in _HNSW_FLAT:
        _read_header(r)
        r.vector(8)   # assign_probas   double
        r.vector(4)   # cum_nneighbor_per_level int
        r.vector(4)   # levels          int
        r.vector(8)   # offsets         size_t
        r.vector(4)   # neighbors       int32
        r.take("<iiiii")  # entry_point, max_level, efConstruction, efSearch, upper_beam
        try:
            return _read_flat(r)
        except ValueError:
            # layout drift between faiss versions: locate the nested flat container by signature
            for sig in (b"IxFI", b"IxF2"):
                pos = data.find(sig, 4)
                if pos >= 0:
                    r.o = pos
                    return _read_flat(r)
            raise
    raise ValueError(f"faiss index: unsupported container fourcc 0x{h:08x} (flat / HNSW-flat only)")


# ---------------------------------------------------------------------------
# *.jsonl metadata
# ---------------------------------------------------------------------------
def write_faiss_meta(path: Path, chunks: Sequence[LawChunk]) -> None:
    """faiss_builder.py:99-104: `c.model_dump_json()` per line."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("w", encoding="utf-8") as f:
        for c in chunks:
            f.write(c.model_dump_json() + "\n")


def read_faiss_meta(path: Path) -> List[LawChunk]:
    out: List[LawChunk] = []
    with Path(path).open("r", encoding="utf-8") as f:
        for line in f:
            if line.strip():
                out.append(LawChunk.model_validate(json.loads(line)))
    return out


def write_colbert_meta(path: Path, chunks: Sequence[LawChunk]) -> None:
    """colbert_builder.py:39-52: {"pid": row, "chunk": model_dump()} per line."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("w", encoding="utf-8") as f:
        for pid, c in enumerate(chunks):
            f.write(json.dumps({"pid": pid, "chunk": c.model_dump()}, ensure_ascii=False) + "\n")


def read_colbert_meta(path: Path) -> Dict[int, LawChunk]:
    out: Dict[int, LawChunk] = {}
    with Path(path).open("r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rec = json.loads(line)
                out[int(rec["pid"])] = LawChunk.model_validate(rec["chunk"])
    return out


# ---------------------------------------------------------------------------
# bm25.pkl
# ---------------------------------------------------------------------------
_SAFE_GLOBALS = {
    ("copyreg", "_reconstructor"), ("copyreg", "__newobj__"), ("copy_reg", "_reconstructor"),
    ("builtins", "object"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "set"),
    ("builtins", "tuple"), ("builtins", "frozenset"), ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "str"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        if (module, name) in (("rank_bm25", "BM25Okapi"), ("legal_rag_amd.bm25_model", "BM25Okapi")):
            return BM25Okapi
        if (module, name) in (("legalrag.schemas", "LawChunk"), ("legal_rag_amd.schemas", "LawChunk")):
            return LawChunk
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"bm25.pkl: global {module}.{name} is not allowed")


def read_bm25_pickle(path: Path) -> Tuple[BM25Okapi, List[LawChunk]]:
    with Path(path).open("rb") as f:
        obj = _RestrictedUnpickler(f).load()
    if not isinstance(obj, dict):
        raise RuntimeError(f"[BM25] invalid index file (not a dict): {path}")
    bm25 = obj.get("bm25")
    if bm25 is None:
        raise RuntimeError(f"[BM25] invalid index file (missing 'bm25'): {path}")
    if not isinstance(bm25, BM25Okapi):
        raise RuntimeError(f"[BM25] unsupported bm25 object in index: {type(bm25)}")
    for attr in ("k1", "b", "corpus_size", "avgdl", "doc_freqs", "idf", "doc_len"):
        if not hasattr(bm25, attr):
            raise RuntimeError(f"[BM25] index object lacks attribute '{attr}': {path}")
    # which segmenter produced the index tokens (written by this build only; a reference-built
    # pickle has no such key: English regex words or jieba, bm25_builder.py:39-44)
    bm25.__dict__["_tokenizer_id"] = obj.get("tokenizer") if isinstance(obj.get("tokenizer"), str) else None
    chunks: List[LawChunk] = []
    for c in obj.get("chunks", []):
        if isinstance(c, LawChunk):
            chunks.append(c)
        elif isinstance(c, dict):
            chunks.append(LawChunk(**c))
        else:
            raise RuntimeError(f"[BM25] unsupported chunk format in index: {type(c)}")
    return bm25, chunks


class _BM25Pickler(pickle.Pickler):
    """Emits BM25Okapi instances as GLOBAL rank_bm25.BM25Okapi + state dict."""

    def __init__(self, f, proxy):
        super().__init__(f, protocol=4)
        self._proxy = proxy

    def reducer_override(self, obj):
        if isinstance(obj, BM25Okapi):
            import copyreg
            # object.__new__(rank_bm25.BM25Okapi) + __dict__.update(state) on load
            return copyreg._reconstructor, (self._proxy, object, None), obj.__getstate__()
        return NotImplemented


def write_bm25_pickle(path: Path, bm25: BM25Okapi, chunks: Sequence[LawChunk], tokenizer: Optional[str] = None) -> None:
    """bm25_builder.py:46-51 payload, atomically replaced
    (incremental_bm25_builder.py:76-79).  `tokenizer` (id of the segmenter that produced the
    index tokens, text.tokenizer_id) is stored as a third key the reference's loader ignores."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    real = sys.modules.get("rank_bm25")
    shim = None
    try:
        import rank_bm25 as real_mod  # noqa: F401
        proxy = real_mod.BM25Okapi
    except Exception:  # noqa: BLE001 - wheel absent: name the class without importing it
        shim = types.ModuleType("rank_bm25")
        proxy = type("BM25Okapi", (object,), {"__module__": "rank_bm25"})
        shim.BM25Okapi = proxy
        sys.modules["rank_bm25"] = shim
    try:
        buf = io.BytesIO()
        payload = {"bm25": bm25, "chunks": [c.model_dump() for c in chunks]}
        if tokenizer:
            payload["tokenizer"] = str(tokenizer)
        _BM25Pickler(buf, proxy).dump(payload)
    finally:
        if shim is not None:
            if real is not None:
                sys.modules["rank_bm25"] = real
            else:
                sys.modules.pop("rank_bm25", None)
    tmp = path.with_suffix(path.suffix + ".tmp")
    tmp.write_bytes(buf.getvalue())
    tmp.replace(path)


# ---------------------------------------------------------------------------
# ColBERT token store (this build's own file inside the ColBERT index directory)
# ---------------------------------------------------------------------------
def colbert_index_dir(index_path: str, experiment: str, index_name: str) -> Path:
    """colbert-ai's layout: <root>/<experiment>/indexes/<name>/ (colbert_builder.py:120-134)."""
    return Path(index_path) / experiment / "indexes" / index_name


def write_token_store(dirpath: Path, D: np.ndarray, doc_ptr: np.ndarray) -> Path:
    dirpath = Path(dirpath)
    dirpath.mkdir(parents=True, exist_ok=True)
    out = dirpath / "amdr_tokens.npz"
    tmp = dirpath / "amdr_tokens.tmp.npz"
    np.savez(tmp, D=np.ascontiguousarray(D, dtype=np.float32), doc_ptr=np.ascontiguousarray(doc_ptr, dtype=np.int64))
    tmp.replace(out)
    return out


def read_token_store(dirpath: Path) -> Tuple[np.ndarray, np.ndarray]:
    """(token embeddings fp32 [tokens, dim], doc_ptr i64 [n_docs + 1]) of a ColBERT index directory:
    this build's own store if present, else a colbert-ai (PLAID) index decompressed on load."""
    p = Path(dirpath) / "amdr_tokens.npz"
    if p.exists():
        z = np.load(p)
        return z["D"], z["doc_ptr"]
    if is_plaid_index(dirpath):
        return read_plaid_index(dirpath)
    raise RuntimeError(f"ColBERT token store not found: {p} (and {dirpath} is not a colbert-ai index directory). "
                       f"Run build_colbert_index() first.")


# ---------------------------------------------------------------------------
# colbert-ai (PLAID) index directory  [upstream layout, from memory — verify]
#   metadata.json            {"config": {"nbits", "dim", ...}, "num_chunks", "num_partitions", "num_embeddings", ...}
#   centroids.pt             half [num_partitions, dim]
#   buckets.pt               (bucket_cutoffs [2^nbits - 1], bucket_weights [2^nbits])
#   avg_residual.pt          scalar (not needed to decompress)
#   <c>.codes.pt             int32 [n_c]            centroid id per token embedding of chunk c
#   <c>.residuals.pt         uint8 [n_c, dim * nbits / 8]   packed bucket indices
#   <c>.metadata.json        {"passage_offset", "num_passages", "num_embeddings", "embedding_offset"}
#   doclens.<c>.json         [tokens of each passage of chunk c]
#   ivf.pid.pt               inverted lists centroid -> pids (candidate generation only: not needed,
#                            this build scores every document exactly)
# Codec (ResidualCodec.compress / binarize / decompress): per dimension the residual
# (embedding - centroid) is bucketized into 2^nbits buckets; the index's bits are written LSB
# FIRST and the flat bit string is packed with np.packbits (first bit -> MSB of the byte).
# decompress = normalize(centroid + bucket_weights[index]).
# ---------------------------------------------------------------------------
def is_plaid_index(dirpath: Path) -> bool:
    d = Path(dirpath)
    return (d / "metadata.json").exists() and (d / "centroids.pt").exists() and (d / "0.codes.pt").exists()


def _torch_load(path: Path):
    import torch
    return torch.load(str(path), map_location="cpu", weights_only=True)


def plaid_unpack_indices(packed: np.ndarray, nbits: int, dim: int) -> np.ndarray:
    """uint8 [n, dim * nbits / 8] -> bucket index per dimension, uint8 [n, dim]."""
    n = packed.shape[0]
    bits = np.unpackbits(np.ascontiguousarray(packed, dtype=np.uint8), axis=1)  # MSB first = write order
    bits = bits.reshape(n, dim, nbits)                                            # bit j of the index at position j
    weights = (1 << np.arange(nbits, dtype=np.uint8)).astype(np.uint8)            # LSB first
    return (bits * weights).sum(axis=2).astype(np.uint8)


def read_plaid_index(dirpath: Path) -> Tuple[np.ndarray, np.ndarray]:
    d = Path(dirpath)
    meta = json.loads((d / "metadata.json").read_text())
    cfg = meta.get("config", {})
    nbits = int(cfg.get("nbits", 4))
    if nbits not in (1, 2, 4, 8):
        raise RuntimeError(f"colbert index {d}: unsupported nbits={nbits}")
    centroids = _torch_load(d / "centroids.pt").float().numpy()
    dim = int(centroids.shape[1])
    if int(cfg.get("dim", dim)) != dim:
        raise RuntimeError(f"colbert index {d}: metadata dim {cfg.get('dim')} != centroid dim {dim}")
    buckets = _torch_load(d / "buckets.pt")
    weights = np.asarray(buckets[1].float().numpy(), dtype=np.float32)
    if weights.shape[0] != (1 << nbits):
        raise RuntimeError(f"colbert index {d}: {weights.shape[0]} bucket weights for nbits={nbits}")
    parts, doclens = [], []
    for c in range(int(meta["num_chunks"])):
        codes = _torch_load(d / f"{c}.codes.pt").numpy().astype(np.int64)
        packed = _torch_load(d / f"{c}.residuals.pt").numpy()
        lens = json.loads((d / f"doclens.{c}.json").read_text())
        if packed.shape[0] != codes.shape[0] or packed.shape[1] * 8 != dim * nbits or sum(lens) != codes.shape[0]:
            raise RuntimeError(f"colbert index {d}: chunk {c} is inconsistent "
                               f"(codes {codes.shape}, residuals {packed.shape}, doclens sum {sum(lens)})")
        emb = centroids[codes] + weights[plaid_unpack_indices(packed, nbits, dim)]
        emb /= np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
        parts.append(emb.astype(np.float32))
        doclens.extend(int(x) for x in lens)
    D = np.concatenate(parts, axis=0) if parts else np.zeros((0, dim), dtype=np.float32)
    if "num_embeddings" in meta and int(meta["num_embeddings"]) != D.shape[0]:
        raise RuntimeError(f"colbert index {d}: {D.shape[0]} embeddings read, metadata says {meta['num_embeddings']}")
    doc_ptr = np.concatenate([[0], np.cumsum(doclens)]).astype(np.int64)
    return D, doc_ptr
