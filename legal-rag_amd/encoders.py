"""Text encoders of the retrieval path (host + PyTorch-ROCm plumbing).

The reference calls FlagEmbedding.FlagModel (vector_store.py:65-77,131-155),
colbert-ai's checkpoint encoder and sentence_transformers.CrossEncoder
(rerankers.py:93-116).  None of those wheels, and no model weights, exist in the
build container, so:
  * `TransformersBGE` runs the published BGE recipe on PyTorch-ROCm with plain
    `transformers` (CLS pooling, L2 normalisation, fp16 on GPU, query
    instruction prepended) when a LOCAL checkpoint directory is available;
    encoder-output parity with FlagModel is unpinned (no weights to compare).
  * `HashingEmbedder` / `HashingTokenEmbedder` / `HashingCrossScorer` are
    deterministic stand-ins (seeded hashed-token random projections) used by
    tests and bench.py so the ENGINE can be exercised end to end.  They are
    only ever selected explicitly (`encoder_backend="hashing"`), never as a
    silent fallback.
"""
from __future__ import annotations

import hashlib
import os
import re
from typing import List, Optional, Sequence

import numpy as np

QUERY_INSTRUCTION = "为这个法律问题生成表示以用于检索相关法律条文："  # vector_store.py:72 (used for zh AND en)

_WORD_RE = re.compile(r"[A-Za-z0-9]+|[一-鿿]")


def _tok(text: str) -> List[str]:
    return _WORD_RE.findall(text.lower())


def _seed(token: str, salt: str) -> int:
    return int.from_bytes(hashlib.blake2b((salt + "\0" + token).encode("utf-8"), digest_size=8).digest(), "little")


class _VecCache:
    def __init__(self, dim: int, salt: str):
        self.dim, self.salt, self.cache = dim, salt, {}

    def vec(self, token: str) -> np.ndarray:
        v = self.cache.get(token)
        if v is None:
            v = np.random.default_rng(_seed(token, self.salt)).standard_normal(self.dim).astype(np.float32)
            self.cache[token] = v
        return v


class HashingEmbedder:
    """Deterministic bag-of-hashed-tokens sentence embedding (stand-in for BGE)."""

    def __init__(self, dim: int = 768, salt: str = "bge-standin"):
        self.dim = dim
        self._c = _VecCache(dim, salt)

    @property
    def hidden_size(self) -> int:
        return self.dim

    def _one(self, text: str) -> np.ndarray:
        toks = _tok(text)
        acc = np.zeros(self.dim, dtype=np.float64)
        if not toks:
            toks = ["<empty>"]
        counts = {}
        for t in toks:
            counts[t] = counts.get(t, 0) + 1
        for t, c in counts.items():
            acc += (1.0 + np.log(c)) * self._c.vec(t)
        n = np.linalg.norm(acc)
        return (acc / (n if n > 0 else 1.0)).astype(np.float32)

    def encode(self, texts: Sequence[str], batch_size: int = 64, max_length: int = 512) -> np.ndarray:
        if isinstance(texts, str):
            return self._one(texts)
        if len(texts) == 0:
            return np.zeros((0, self.dim), dtype=np.float32)
        return np.stack([self._one(t) for t in texts])

    def encode_queries(self, texts: Sequence[str], batch_size: int = 64, max_length: int = 512) -> np.ndarray:
        # the stand-in ignores the instruction prefix (it would add the same vector to every query)
        return self.encode(texts, batch_size=batch_size, max_length=max_length)


class HashingTokenEmbedder:
    """Stand-in for the ColBERT checkpoint: one seeded unit 128-d vector per token,
    mixed with a document-position-free context vector.  Queries are padded to
    32 tokens with a [MASK] vector like ColBERT's query augmentation."""

    def __init__(self, dim: int = 128, doc_maxlen: int = 220, query_maxlen: int = 32, salt: str = "colbert-standin"):
        self.dim, self.doc_maxlen, self.query_maxlen = dim, doc_maxlen, query_maxlen
        self._c = _VecCache(dim, salt)

    def _unit(self, toks: List[str]) -> np.ndarray:
        M = np.stack([self._c.vec(t) for t in toks]).astype(np.float32)
        M /= np.linalg.norm(M, axis=1, keepdims=True)
        return M

    def encode_doc(self, text: str) -> np.ndarray:
        toks = (["[D]"] + _tok(text))[: self.doc_maxlen]
        return self._unit(toks)

    def encode_query(self, text: str) -> np.ndarray:
        toks = (["[Q]"] + _tok(text))[: self.query_maxlen]
        toks = toks + ["[MASK]"] * (self.query_maxlen - len(toks))
        return self._unit(toks)

    def encode_queries(self, texts: Sequence[str]) -> np.ndarray:
        """[n, query_maxlen, dim]: encode_query of every text, one gather + one normalisation for the batch (row-wise
        arithmetic: the same bits as the per-query call)."""
        rows = []
        for text in texts:
            toks = (["[Q]"] + _tok(text))[: self.query_maxlen]
            rows.extend(toks)
            rows.extend(["[MASK]"] * (self.query_maxlen - len(toks)))
        if not rows:
            return np.zeros((0, self.query_maxlen, self.dim), dtype=np.float32)
        return self._unit(rows).reshape(len(texts), self.query_maxlen, self.dim)


class HashingCrossScorer:
    """Stand-in for the cross-encoder: sigmoid of a token-overlap statistic."""

    _DOC_CACHE_MAX = 1 << 16

    def __init__(self):
        self._docs = {}  # text -> (token counts, token total): a corpus chunk is scored against many queries

    def _doc(self, d: str):
        ent = self._docs.get(d)
        if ent is None:
            t = _tok(d)
            counts = {}
            for x in t:
                counts[x] = counts.get(x, 0) + 1
            ent = (counts, len(t))
            if len(self._docs) >= self._DOC_CACHE_MAX:
                self._docs.clear()
            self._docs[d] = ent
        return ent

    def score_batch(self, query: str, docs: List[str]) -> List[float]:
        q = set(_tok(query))
        out = []
        for d in docs:
            head = getattr(d, "head", None)
            if head is not None and head[-1:] == ")" and d.tail[:1] == " ":
                # hybrid_retriever.HitText: the chunk part is shared by every hit of that chunk; the two parts meet at
                # ") " — no token spans them, so the tokens of the whole are the tokens of the parts
                counts, total = self._doc(head)
                t = _tok(d.tail)
                hit = sum(counts.get(x, 0) for x in q) + sum(1 for x in t if x in q)
                total += len(t)
            else:
                counts, total = self._doc(d)
                hit = sum(counts.get(x, 0) for x in q)
            # `hit` = document tokens that occur in the query, with multiplicity (an integer: counted from either side)
            ov = hit / (1.0 + total) if total else 0.0
            h = (_seed(d, query) % 1000) / 1e6  # tiny deterministic jitter: no exact ties
            out.append(float(1.0 / (1.0 + np.exp(-(8.0 * ov - 1.0))) + h))
        return out

    def score(self, query: str, doc: str) -> float:
        return self.score_batch(query, [doc])[0]

    def score_pairs(self, pairs) -> List[float]:
        return [self.score_batch(q, [d])[0] for q, d in pairs]


class TransformersBGE:
    """BGE sentence encoder on PyTorch-ROCm (plain transformers; FlagModel recipe)."""

    def __init__(self, model_path: str, device: Optional[str] = None,
                 query_instruction: str = QUERY_INSTRUCTION):
        import torch
        from transformers import AutoModel, AutoTokenizer

        self.torch = torch
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.tokenizer = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
        self.model = AutoModel.from_pretrained(model_path, local_files_only=True).to(self.device).eval()
        self.use_fp16 = self.device.type == "cuda"
        if self.use_fp16:
            self.model = self.model.half()
        self.query_instruction = query_instruction

    @property
    def hidden_size(self) -> int:
        return int(self.model.config.hidden_size)

    def encode_tensor(self, texts: Sequence[str], batch_size: int = 64, max_length: int = 512, is_query: bool = False):
        """fp32 [n, hidden] tensor ON `self.device`, L2-normalised: the embedding goes from the
        encoder to the dense kernel (amdr_dense_search_device) without a host hop."""
        torch = self.torch
        texts = [self.query_instruction + t for t in texts] if is_query else list(texts)
        out = torch.zeros((len(texts), self.hidden_size), dtype=torch.float32, device=self.device)
        if not texts:
            return out
        order = np.argsort([-len(t) for t in texts])  # length-sorted batches, like FlagModel
        with torch.inference_mode():
            for s in range(0, len(texts), batch_size):
                idx = order[s:s + batch_size]
                enc = self.tokenizer([texts[i] for i in idx], padding=True, truncation=True, max_length=max_length,
                                     return_tensors="pt").to(self.device)
                h = self.model(**enc).last_hidden_state[:, 0]
                out[torch.as_tensor(idx.copy(), device=self.device)] = torch.nn.functional.normalize(h.float(), dim=-1)
        return out

    def encode(self, texts, batch_size: int = 64, max_length: int = 512) -> np.ndarray:
        single = isinstance(texts, str)
        if single:
            texts = [texts]
        if len(texts) == 0:
            return np.zeros((0, self.hidden_size), dtype=np.float32)
        out = self.encode_tensor(texts, batch_size, max_length).cpu().numpy()
        return out[0] if single else out

    def encode_queries(self, texts, batch_size: int = 64, max_length: int = 512) -> np.ndarray:
        if isinstance(texts, str):
            return self.encode(self.query_instruction + texts, batch_size, max_length)
        return self.encode([self.query_instruction + t for t in texts], batch_size, max_length)


class TransformersColBERT:
    """ColBERT token encoder for BERT-style checkpoints (colbert-ir/colbertv2.0 layout:
    a BERT encoder + a bias-free `linear.weight` [dim, hidden]) on PyTorch-ROCm, following
    the published ColBERT recipe: query = [CLS] [Q] tokens, padded to `query_maxlen` with
    [MASK] (query augmentation; the pads are not attended to but their outputs are kept);
    document = [CLS] [D] tokens truncated to `doc_maxlen`, punctuation tokens dropped;
    every token vector L2-normalised.  [Q]/[D] are the tokenizer's [unused0]/[unused1].
    Parity with colbert-ai is unpinned (the wheel and the weights are absent here)."""

    def __init__(self, model_path: str, doc_maxlen: int = 220, query_maxlen: int = 32, device: Optional[str] = None):
        import string

        import torch
        from safetensors import safe_open
        from transformers import AutoModel, AutoTokenizer

        self.torch = torch
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.tok = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
        self.bert = AutoModel.from_pretrained(model_path, local_files_only=True).to(self.device).eval()
        w = None
        st = os.path.join(model_path, "model.safetensors")
        if os.path.exists(st):
            with safe_open(st, framework="pt") as f:
                for key in f.keys():
                    if key.endswith("linear.weight"):
                        w = f.get_tensor(key)
        lin = os.path.join(model_path, "colbert_linear.safetensors")
        if w is None and os.path.exists(lin):
            with safe_open(lin, framework="pt") as f:
                w = f.get_tensor("linear.weight")
        if w is None:
            raise RuntimeError(f"no ColBERT projection (linear.weight) found under {model_path}")
        self.linear = w.to(self.device).float()
        self.dim = int(self.linear.shape[0])
        self.doc_maxlen, self.query_maxlen = doc_maxlen, query_maxlen
        self.q_id = self.tok.convert_tokens_to_ids("[unused0]")
        self.d_id = self.tok.convert_tokens_to_ids("[unused1]")
        self.skip = {self.tok.convert_tokens_to_ids(c) for c in string.punctuation}
        self.skip.discard(self.tok.unk_token_id)

    def _forward(self, ids, mask):
        torch = self.torch
        with torch.inference_mode():
            h = self.bert(input_ids=ids, attention_mask=mask).last_hidden_state.float()
            v = h @ self.linear.T
            return torch.nn.functional.normalize(v, dim=-1)

    def encode_query(self, text: str) -> np.ndarray:
        torch = self.torch
        body = self.tok(text, add_special_tokens=False)["input_ids"][: self.query_maxlen - 3]
        ids = [self.tok.cls_token_id, self.q_id] + body + [self.tok.sep_token_id]
        n_real = len(ids)
        ids = ids + [self.tok.mask_token_id] * (self.query_maxlen - n_real)
        mask = [1] * n_real + [0] * (self.query_maxlen - n_real)
        out = self._forward(torch.tensor([ids], device=self.device), torch.tensor([mask], device=self.device))
        return out[0].cpu().numpy().astype(np.float32)

    def encode_queries_tensor(self, texts: Sequence[str]):
        """Device tensor [n, query_maxlen, dim] of a batch of queries in ONE forward: every query is padded to
        query_maxlen with [MASK] anyway (query augmentation), so the batch is rectangular by construction.  The rows equal
        encode_query's up to the batched GEMMs' rounding."""
        torch = self.torch
        ids_b, mask_b = [], []
        for text in texts:
            body = self.tok(text, add_special_tokens=False)["input_ids"][: self.query_maxlen - 3]
            ids = [self.tok.cls_token_id, self.q_id] + body + [self.tok.sep_token_id]
            n_real = len(ids)
            ids_b.append(ids + [self.tok.mask_token_id] * (self.query_maxlen - n_real))
            mask_b.append([1] * n_real + [0] * (self.query_maxlen - n_real))
        if not ids_b:
            return torch.zeros((0, self.query_maxlen, self.dim), device=self.device)
        return self._forward(torch.tensor(ids_b, device=self.device), torch.tensor(mask_b, device=self.device)).contiguous()

    def encode_queries(self, texts: Sequence[str]) -> np.ndarray:
        return self.encode_queries_tensor(texts).cpu().numpy().astype(np.float32)

    def encode_doc(self, text: str) -> np.ndarray:
        torch = self.torch
        body = self.tok(text, add_special_tokens=False)["input_ids"][: self.doc_maxlen - 3]
        ids = [self.tok.cls_token_id, self.d_id] + body + [self.tok.sep_token_id]
        out = self._forward(torch.tensor([ids], device=self.device), torch.ones((1, len(ids)), dtype=torch.long,
                                                                               device=self.device))[0]
        keep = [j for j, t in enumerate(ids) if t not in self.skip]
        return out[keep].cpu().numpy().astype(np.float32)


_EMBEDDER_CACHE = {}


def get_embedder(model_name: str, backend: str = "auto", dim: int = 768, device: Optional[str] = None):
    """Resolve the sentence encoder.  'hashing' -> stand-in; 'transformers' ->
    local checkpoint required; 'auto' -> local checkpoint if `model_name` is a
    directory, else an error naming what is missing (no silent stand-in)."""
    key = (model_name, backend, dim, device)
    if key in _EMBEDDER_CACHE:
        return _EMBEDDER_CACHE[key]
    if backend == "hashing":
        m = HashingEmbedder(dim=dim)
    elif os.path.isdir(model_name):
        m = TransformersBGE(model_name, device=device)  # cfg.retrieval.device, not whatever "cuda" defaults to
    else:
        raise RuntimeError(
            f"embedding model '{model_name}' is not a local checkpoint directory and this build has no network; "
            f"point cfg.retrieval.embedding_model at a downloaded BGE checkpoint or set "
            f"cfg.retrieval.encoder_backend='hashing' to use the deterministic stand-in.")
    _EMBEDDER_CACHE[key] = m
    return m
