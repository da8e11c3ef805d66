"""The encoder leg on the MI355X (vector_store.py:65-77,131-155, rerankers.py:93-116): the same tiny
randomly initialised checkpoints as tests/test_encoders.py, forward passes on the device in the
precision the product uses there, against the fp32 CPU forward.  Real BGE / ColBERT / reranker
weights are not available offline: this pins the recipe and the device plumbing, not a model."""
import numpy as np
import pytest
import torch

from test_encoders import tiny  # noqa: F401  (module-scoped fixture: tiny BERT checkpoints)

from legal_rag_amd import encoders
from legal_rag_amd.retrieval import rerankers

pytestmark = pytest.mark.gpu

TOL_FP16 = 2e-2  # fp16 forward on the device vs fp32 on the CPU, unit-norm outputs

TEXTS = ["the seller goods", "buyer", "warranty of the merchant goods the goods", "", "of the the of goods seller"]


def test_bge_on_device_fp16_matches_cpu_fp32(tiny):  # noqa: F811
    gpu = encoders.TransformersBGE(tiny["bge"], device="cuda:0")
    cpu = encoders.TransformersBGE(tiny["bge"], device="cpu")
    assert gpu.use_fp16 and next(gpu.model.parameters()).dtype == torch.float16  # FlagModel: fp16 when CUDA
    E, R = gpu.encode(TEXTS, batch_size=2), cpu.encode(TEXTS, batch_size=2)
    assert E.shape == R.shape == (5, 32) and E.dtype == np.float32
    assert np.allclose(np.linalg.norm(E, axis=1), 1.0, atol=1e-3)       # L2-normalised in fp32
    assert np.max(np.abs(E - R)) <= TOL_FP16                              # CLS pooling, same recipe
    q, rq = gpu.encode_queries(["the goods"]), cpu.encode_queries(["the goods"])
    assert np.max(np.abs(q - rq)) <= TOL_FP16
    assert np.max(np.abs(q - gpu.encode([encoders.QUERY_INSTRUCTION + "the goods"]))) <= 1e-6  # instruction prepended


def test_embedding_goes_to_the_dense_kernel_without_a_host_hop(tiny):  # noqa: F811
    from legal_rag_amd import _native
    gpu = encoders.TransformersBGE(tiny["bge"], device="cuda:0")
    docs = [f"{a} {b} goods" for a in ("the", "seller", "buyer", "warranty", "merchant") for b in ("of", "the", "goods")]
    X = gpu.encode_tensor(docs)
    q = gpu.encode_tensor(["seller goods", "the warranty"], is_query=True)
    assert X.is_cuda and q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=X.shape[0], dim=32, device=0, keepalive=X)
    s = torch.empty((2, 5), dtype=torch.float32, device="cuda:0")
    i = torch.empty((2, 5), dtype=torch.int64, device="cuda:0")
    idx.search_device(q.data_ptr(), 2, 5, s.data_ptr(), i.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    hs, hi = idx.search(q.cpu().numpy(), 5)                                # same vectors through the host API
    assert np.array_equal(i.cpu().numpy(), hi) and np.array_equal(s.cpu().numpy(), hs)
    ref = (q @ X.T).cpu().numpy()
    assert np.max(np.abs(np.take_along_axis(ref, hi, axis=1) - hs)) <= 1e-5
    idx.close()


def test_vector_store_embed_device(tiny, tmp_path):  # noqa: F811
    from legal_rag_amd.config import AppConfig
    from legal_rag_amd.retrieval.vector_store import VectorStore
    cfg = AppConfig.for_data_dir(str(tmp_path), "en")
    cfg.retrieval.embedding_model = tiny["bge"]
    cfg.retrieval.encoder_backend = "auto"
    vs = VectorStore(cfg)
    assert isinstance(vs.model, encoders.TransformersBGE) and str(vs.model.device) == "cuda:0"  # cfg.retrieval.device
    t = vs.embed_device(["the goods", "buyer"], is_query=True)
    assert t.is_cuda and t.shape == (2, 32)
    assert np.max(np.abs(t.cpu().numpy() - vs._embed(["the goods", "buyer"], is_query=True))) <= 1e-6
    cfg.retrieval.encoder_backend = "hashing"
    vh = VectorStore(cfg)
    th = vh.embed_device(["the goods"], is_query=True)
    assert th.is_cuda and np.array_equal(th.cpu().numpy(), vh._embed(["the goods"], is_query=True))


def test_colbert_on_device(tiny):  # noqa: F811
    gpu = encoders.TransformersColBERT(tiny["bge"], doc_maxlen=24, device="cuda:0")
    cpu = encoders.TransformersColBERT(tiny["bge"], doc_maxlen=24, device="cpu")
    q, rq = gpu.encode_query("the seller?"), cpu.encode_query("the seller?")
    assert q.shape == (32, 16)                                                # padded to 32 with [MASK]
    assert np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-4) and np.max(np.abs(q - rq)) <= TOL_FP16
    d, rd = gpu.encode_doc("the goods, the seller. warranty"), cpu.encode_doc("the goods, the seller. warranty")
    assert d.shape == rd.shape == (8, 16) and np.max(np.abs(d - rd)) <= TOL_FP16  # punctuation dropped
    assert gpu.encode_doc("goods " * 100).shape[0] == 24


def test_cross_encoder_on_device(tiny):  # noqa: F811
    docs = ["the goods", "warranty of the merchant", "buyer"]
    cpu = rerankers.CrossEncoderReranker(model_name=tiny["ce"], device="cpu", batch_size=2)
    f32 = rerankers.CrossEncoderReranker(model_name=tiny["ce"], device="cuda:0", batch_size=2)
    f16 = rerankers.CrossEncoderReranker(model_name=tiny["ce"], device="cuda:0", batch_size=2, fp16=True)
    ref = cpu.score_batch("seller goods", docs)
    got32, got16 = f32.score_batch("seller goods", docs), f16.score_batch("seller goods", docs)
    assert next(f32._model.parameters()).dtype == torch.float32             # CrossEncoder.predict runs fp32
    assert next(f16._model.parameters()).dtype == torch.float16             # explicit knob only
    assert all(0.0 < g < 1.0 for g in got32)                                # sigmoid(logit) of a 1-label head
    assert np.max(np.abs(np.array(got32) - np.array(ref))) <= 1e-4
    assert np.max(np.abs(np.array(got16) - np.array(ref))) <= TOL_FP16
