"""Index artifact readers / writers (layout of SURVEY.md §8 a-12)."""
import io
import json
import pickle
import struct

import numpy as np
import pytest

from legal_rag_amd import artifacts
from legal_rag_amd.bm25_model import BM25Okapi
from legal_rag_amd.schemas import LawChunk


def chunks(n=5):
    return [LawChunk(id=f"f.txt::{i}", law_name="Code", article_no=f"§ {i}", article_id=str(i),
                     text=f"goods sold by merchant number {i}", lang="en", source="f.txt") for i in range(n)]


def test_faiss_flat_roundtrip(tmp_path):
    X = np.random.default_rng(0).standard_normal((37, 768)).astype(np.float32)
    p = tmp_path / "faiss" / "faiss.index"
    artifacts.write_faiss_flat_ip(p, X)
    raw = p.read_bytes()
    assert raw[:4] == b"IxFI"
    d, n = struct.unpack_from("<iq", raw, 4)
    assert (d, n) == (768, 37)
    Y, metric = artifacts.read_faiss_index(p)
    assert metric == artifacts.METRIC_INNER_PRODUCT and np.array_equal(X, Y)


def test_faiss_hnsw_container_storage_is_extracted(tmp_path):
    """Synthesise an IndexHNSWFlat container (what the reference writes) and read its flat storage."""
    X = np.random.default_rng(1).standard_normal((9, 16)).astype(np.float32)
    flat = io.BytesIO()
    flat.write(b"IxFI" + struct.pack("<iqqq?i", 16, 9, 1 << 20, 1 << 20, True, 0))
    flat.write(struct.pack("<Q", X.size) + X.tobytes())

    def vec(fmt, vals):
        return struct.pack("<Q", len(vals)) + struct.pack(f"<{len(vals)}{fmt}", *vals)
    buf = io.BytesIO()
    buf.write(b"IHNf" + struct.pack("<iqqq?i", 16, 9, 1 << 20, 1 << 20, True, 0))
    buf.write(vec("d", [0.9, 0.1]) + vec("i", [0, 64, 96]) + vec("i", [1] * 9) + vec("Q", list(range(10))) +
              vec("i", [-1] * 40))
    buf.write(struct.pack("<iiiii", 3, 1, 200, 128, 1))
    buf.write(flat.getvalue())
    p = tmp_path / "h.index"
    p.write_bytes(buf.getvalue())
    Y, metric = artifacts.read_faiss_index(p)
    assert np.array_equal(X, Y) and metric == 0


def test_faiss_rejects_garbage(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"IxPQ" + b"\0" * 64)
    with pytest.raises(ValueError):
        artifacts.read_faiss_index(p)
    p.write_bytes(b"IxFI" + struct.pack("<iqqq?i", 8, 4, 0, 0, True, 0) + struct.pack("<Q", 1 << 40))
    with pytest.raises(ValueError):
        artifacts.read_faiss_index(p)


def test_meta_jsonl_matches_reference_format(tmp_path):
    cs = chunks(3)
    p = tmp_path / "faiss_meta.jsonl"
    artifacts.write_faiss_meta(p, cs)
    lines = p.read_text(encoding="utf-8").splitlines()
    assert lines == [c.model_dump_json() for c in cs]
    assert list(json.loads(lines[0])) == ["id", "law_name", "chapter", "section", "article_no", "article_id", "text",
                                          "lang", "source", "start_char", "end_char"]
    assert artifacts.read_faiss_meta(p) == cs
    q = tmp_path / "colbert" / "colbert_meta.jsonl"
    artifacts.write_colbert_meta(q, cs)
    rec = json.loads(q.read_text(encoding="utf-8").splitlines()[1])
    assert rec["pid"] == 1 and rec["chunk"]["id"] == "f.txt::1"
    assert artifacts.read_colbert_meta(q)[2] == cs[2]


def test_bm25_pickle_names_rank_bm25_and_roundtrips(tmp_path):
    cs = chunks(6)
    bm = BM25Okapi([c.text.split() for c in cs])
    p = tmp_path / "bm25.pkl"
    artifacts.write_bm25_pickle(p, bm, cs)
    raw = p.read_bytes()
    assert b"rank_bm25" in raw and b"BM25Okapi" in raw and b"legal_rag_amd" not in raw
    import sys
    assert "rank_bm25" not in sys.modules  # the naming shim does not leak
    bm2, cs2 = artifacts.read_bm25_pickle(p)
    assert cs2 == cs
    for attr in ("k1", "b", "epsilon", "corpus_size", "avgdl", "doc_freqs", "idf", "doc_len", "average_idf"):
        assert getattr(bm2, attr) == getattr(bm, attr), attr
    assert list(bm2.idf) == list(bm.idf)  # vocabulary order survives


def test_bm25_pickle_is_restricted(tmp_path):
    class Evil:
        def __reduce__(self):
            import os
            return os.system, ("echo pwned",)
    p = tmp_path / "bm25.pkl"
    p.write_bytes(pickle.dumps({"bm25": Evil(), "chunks": []}))
    with pytest.raises(pickle.UnpicklingError):
        artifacts.read_bm25_pickle(p)
    p.write_bytes(pickle.dumps({"chunks": []}))
    with pytest.raises(RuntimeError):
        artifacts.read_bm25_pickle(p)


def test_token_store_roundtrip(tmp_path):
    D = np.random.default_rng(2).standard_normal((11, 128)).astype(np.float32)
    ptr = np.array([0, 3, 4, 11])
    d = artifacts.colbert_index_dir(str(tmp_path / "colbert"), "experiment", "law_en")
    assert d == tmp_path / "colbert" / "experiment" / "indexes" / "law_en"
    artifacts.write_token_store(d, D, ptr)
    D2, ptr2 = artifacts.read_token_store(d)
    assert np.array_equal(D, D2) and np.array_equal(ptr, ptr2)
    with pytest.raises(RuntimeError):
        artifacts.read_token_store(tmp_path / "nope")
