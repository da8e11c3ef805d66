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


# ---- colbert-ai (PLAID) index directory: read + decompress (layout restated from memory, see artifacts.py) ----
def _write_plaid_fixture(d, D, doclens, nbits=4, n_centroids=16, chunk_docs=3, seed=0):
    """A small index in colbert-ai's on-disk layout, written with colbert-ai's own packing steps
    (bucketize -> bits LSB first -> np.packbits) on torch tensors."""
    import json
    import torch
    rng = np.random.default_rng(seed)
    dim = D.shape[1]
    centroids = rng.standard_normal((n_centroids, dim)).astype(np.float32)
    centroids /= np.linalg.norm(centroids, axis=1, keepdims=True)
    codes = np.argmax(D @ centroids.T, axis=1).astype(np.int32)
    resid = D - centroids[codes]
    q = np.quantile(resid, np.linspace(0, 1, (1 << nbits) + 1))
    cutoffs = q[1:-1].astype(np.float32)
    weights = np.array([(q[i] + q[i + 1]) / 2 for i in range(1 << nbits)], dtype=np.float32)
    idx = torch.bucketize(torch.from_numpy(resid).float(), torch.from_numpy(cutoffs)).to(torch.uint8)
    bits = (idx.unsqueeze(-1) >> torch.arange(0, nbits, dtype=torch.uint8)) & 1          # ResidualCodec.binarize
    packed = np.packbits(np.asarray(bits.contiguous().flatten())).reshape(D.shape[0], dim // 8 * nbits)
    d.mkdir(parents=True, exist_ok=True)
    torch.save(torch.from_numpy(centroids).half(), d / "centroids.pt")
    torch.save((torch.from_numpy(cutoffs), torch.from_numpy(weights)), d / "buckets.pt")
    torch.save(torch.tensor(float(np.abs(resid).mean())), d / "avg_residual.pt")
    ptr = np.concatenate([[0], np.cumsum(doclens)])
    chunks = [list(range(i, min(i + chunk_docs, len(doclens)))) for i in range(0, len(doclens), chunk_docs)]
    for c, docs in enumerate(chunks):
        lo, hi = int(ptr[docs[0]]), int(ptr[docs[-1] + 1])
        torch.save(torch.from_numpy(codes[lo:hi].copy()), d / f"{c}.codes.pt")
        torch.save(torch.from_numpy(packed[lo:hi].copy()), d / f"{c}.residuals.pt")
        (d / f"doclens.{c}.json").write_text(json.dumps([int(doclens[i]) for i in docs]))
        (d / f"{c}.metadata.json").write_text(json.dumps({"passage_offset": docs[0], "num_passages": len(docs),
                                                          "num_embeddings": hi - lo, "embedding_offset": lo}))
    (d / "metadata.json").write_text(json.dumps({"config": {"nbits": nbits, "dim": dim}, "num_chunks": len(chunks),
                                                 "num_partitions": n_centroids, "num_embeddings": int(D.shape[0]),
                                                 "avg_doclen": float(np.mean(doclens))}))
    # what decompression must give: normalize(half-rounded centroid + bucket weight)
    exp = torch.from_numpy(centroids).half().float().numpy()[codes] + weights[idx.numpy()]
    exp /= np.linalg.norm(exp, axis=1, keepdims=True)
    return exp.astype(np.float32)


@pytest.mark.parametrize("nbits", [4, 2, 1])
def test_plaid_index_directory_is_read_and_decompressed(tmp_path, nbits):
    from legal_rag_amd import artifacts
    rng = np.random.default_rng(3)
    doclens = [5, 1, 17, 220, 3, 8, 40]
    D = rng.standard_normal((sum(doclens), 128)).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    d = artifacts.colbert_index_dir(str(tmp_path / "colbert"), "experiment", "law_en")
    exp = _write_plaid_fixture(d, D, doclens, nbits=nbits)
    assert artifacts.is_plaid_index(d)
    got, doc_ptr = artifacts.read_token_store(d)            # no amdr_tokens.npz: the PLAID files are used
    assert got.dtype == np.float32 and got.shape == D.shape
    assert doc_ptr.tolist() == np.concatenate([[0], np.cumsum(doclens)]).tolist()
    assert np.allclose(got, exp, atol=1e-6)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    if nbits == 4:
        assert float(np.mean(np.sum(got * D, axis=1))) > 0.8  # the codec approximates the embeddings it stored
    # a truncated chunk is refused, not mis-read
    import json
    (d / "doclens.0.json").write_text(json.dumps([5, 1]))
    with pytest.raises(RuntimeError):
        artifacts.read_plaid_index(d)


def test_plaid_unpack_bit_order():
    from legal_rag_amd import artifacts
    # nbits = 4, two dims per byte: index 0b0001 (=1) then 0b0110 (=6), bits written LSB first, packed MSB first
    # dim0 bits 1,0,0,0 ; dim1 bits 0,1,1,0  ->  byte 0b1000_0110 = 0x86
    assert artifacts.plaid_unpack_indices(np.array([[0x86]], dtype=np.uint8), 4, 2).tolist() == [[1, 6]]
    assert artifacts.plaid_unpack_indices(np.array([[0b10110000]], dtype=np.uint8), 2, 4).tolist() == [[1, 3, 0, 0]]
