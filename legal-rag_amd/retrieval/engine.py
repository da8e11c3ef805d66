"""Batched, device-resident hybrid pipeline (the throughput path).

The reference answers one query per `HybridRetriever.search` call
(hybrid_retriever.py:282-384).  This engine runs the same stages — dense top-k,
BM25 top-k, optional MaxSim top-k, `_fuse`, the min_final_score filter and the
optional rerank blend — for a whole batch of queries without leaving HBM:
every stage is a libamdretrieval kernel launched on the caller's stream, and
the buffers between stages are plain device allocations (torch tensors are
used only as the allocator / stream provider).  `HybridRetriever.search_batch`
and bench.py sit on top of it; the single-query API uses the same kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import torch

from .. import _native


@dataclass
class BatchResult:
    ids: torch.Tensor     # i64 [nq, max_out]  fused rank order, -1 padded
    vals: torch.Tensor    # f64 [nq, max_out, 9]  (_native.FV layout)
    mask: torch.Tensor    # i32 [nq, max_out]  channel membership bits
    count: torch.Tensor   # i32 [nq]  hits surviving min_final_score
    dense_ids: Optional[torch.Tensor] = None
    dense_scores: Optional[torch.Tensor] = None
    bm25_ids: Optional[torch.Tensor] = None
    bm25_scores: Optional[torch.Tensor] = None
    colbert_ids: Optional[torch.Tensor] = None
    colbert_scores: Optional[torch.Tensor] = None
    rerank: Optional[torch.Tensor] = None  # f64 [nq, max_out, 2] (raw, norm) after rerank_blend
    packed: Optional[torch.Tensor] = None  # u8: ids | vals | mask | count in ONE allocation (one D2H for the host API)

    def to_host(self):
        """(ids, vals, mask, count) as numpy arrays through ONE device-to-host copy (the first `.cpu()` of a result
        waits for the kernels; three more copies of a single query's few hundred bytes cost ~10 us each)."""
        if self.packed is None:
            return (self.ids.cpu().numpy(), self.vals.cpu().numpy(), self.mask.cpu().numpy(), self.count.cpu().numpy())
        nq, mo = self.ids.shape
        h = self.packed.cpu().numpy()
        o1 = nq * mo * 8
        o2 = o1 + nq * mo * _native.FUSE_NVALS * 8
        o3 = o2 + nq * mo * 4
        return (h[:o1].view("int64").reshape(nq, mo), h[o1:o2].view("float64").reshape(nq, mo, _native.FUSE_NVALS),
                h[o2:o3].view("int32").reshape(nq, mo), h[o3:o3 + nq * 4].view("int32"))


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class HybridEngine:
    def __init__(self, dense: Optional[_native.DenseIndex], bm25: Optional[_native.BM25Index],
                 maxsim: Optional[_native.MaxSimIndex] = None, *, device: int = 0,
                 dense_row2uid: Optional[torch.Tensor] = None, bm25_row2uid: Optional[torch.Tensor] = None,
                 colbert_row2uid: Optional[torch.Tensor] = None, shard_offset: Optional[int] = None,
                 shard_group=None):
        if not torch.cuda.is_available():
            raise RuntimeError("HybridEngine needs a GPU (no CPU fallback)")
        self.dense, self.bm25, self.maxsim = dense, bm25, maxsim
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        self.maps = (dense_row2uid, bm25_row2uid, colbert_row2uid)
        self._bufs = {}
        self._xcache = {}  # argument blocks of the shard exchange (sharding.exchange_topk_native)
        # Row-sharded corpus (retrieval/sharding.py): the three indexes hold this rank's row block (local ids
        # 0 .. n_r - 1 = global ids shard_offset ..); search_batch then all-gathers the packed per-channel lists
        # ONCE per batch, merges W*k -> k per channel (merge_parts_kernel) and fuses the GLOBAL lists — the
        # result is identical on every rank.  None: one index holds everything, nothing is exchanged.
        self.shard_offset, self.shard_group = shard_offset, shard_group

    def _buf(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.tdev)
            self._bufs[key] = t
        return t

    def _pinned(self, name, nbytes: int):
        """A page-locked host buffer of at least nbytes (grown geometrically, reused): staging for the one H2D copy of a
        batch's BM25 query CSR and the one D2H copy of its results — copies from / to pageable memory go through the
        driver's own bounce buffer synchronously."""
        key = ("pin", name)
        t = self._bufs.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty((max(int(nbytes), 2 * (t.numel() if t is not None else 0), 4096),), dtype=torch.uint8).pin_memory()
            self._bufs[key] = t
        return t

    def upload_csr(self, q_ptr, q_terms):
        """BM25 query CSR (numpy int64 [n+1], int32 [total]) -> device tensors through ONE pinned staging copy."""
        import numpy as np
        qp8, qt8 = q_ptr.view(np.uint8), q_terms.view(np.uint8)
        n1, n2 = qp8.size, qt8.size
        host = self._pinned("csr", n1 + n2)
        hv = host.numpy()
        hv[:n1] = qp8
        hv[n1:n1 + n2] = qt8
        dev = self._bufs.get(("csr_dev",))
        if dev is None or dev.numel() < n1 + n2:
            dev = torch.empty((host.numel(),), dtype=torch.uint8, device=self.tdev)
            self._bufs[("csr_dev",)] = dev
        dev[: n1 + n2].copy_(host[: n1 + n2], non_blocking=True)
        return dev[:n1].view(torch.int64), dev[n1:n1 + n2].view(torch.int32)

    def compact_to_host(self, res: "BatchResult", w: int):
        """(rows i64 [nq, w], scores f64 [nq, w], channel mask i32 [nq, w], count i32 [nq]) of a fused result on the host:
        compacted by ONE kernel (amdr_fuse_compact_device), ONE copy into pinned memory, one synchronise."""
        nq, mo = res.ids.shape
        w = max(1, min(int(w), int(mo)))
        o1 = nq * w * 8
        o2 = o1 + nq * w * 8
        o3 = o2 + nq * w * 4
        tot = o3 + nq * 4
        pk = self._buf("cpk", (tot,), torch.uint8)
        base = pk.data_ptr()
        _native.fuse_compact_device(nq, mo, w, res.ids.data_ptr(), res.vals.data_ptr(), res.mask.data_ptr(),
                                    res.count.data_ptr(), base, base + o1, base + o2, base + o3, device=self.device,
                                    stream=_stream())
        host = self._pinned("cpkh", tot)
        host[:tot].copy_(pk, non_blocking=True)
        torch.cuda.current_stream(self.tdev).synchronize()
        h = host.numpy()
        return (h[:o1].view("int64").reshape(nq, w).copy(), h[o1:o2].view("float64").reshape(nq, w).copy(),
                h[o2:o3].view("int32").reshape(nq, w).copy(), h[o3:tot].view("int32").copy())

    def reserve(self, nq: int, k: int, total_terms: int = 0) -> None:
        if self.dense is not None:
            self.dense.reserve(nq, k)
        if self.bm25 is not None:
            self.bm25.reserve(nq, k, max(total_terms, 1))
        if self.maxsim is not None:
            self.maxsim.reserve(nq, k)

    # -- channels (device in, device out) -----------------------------------
    def dense_topk(self, q_emb: torch.Tensor, k: int):
        nq = q_emb.shape[0]
        assert q_emb.is_cuda and q_emb.dtype == torch.float32 and q_emb.is_contiguous()
        s = self._buf("ds", (nq, k), torch.float32)
        i = self._buf("di", (nq, k), torch.int64)
        self.dense.search_device(q_emb.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), _stream())
        return s, i

    def bm25_topk(self, q_terms: torch.Tensor, q_ptr: torch.Tensor, k: int):
        nq = q_ptr.shape[0] - 1
        assert q_terms.dtype == torch.int32 and q_ptr.dtype == torch.int64 and q_terms.is_cuda and q_ptr.is_cuda
        s = self._buf("bs", (nq, k), torch.float64)
        i = self._buf("bi", (nq, k), torch.int64)
        self.bm25.search_device(q_terms.data_ptr(), q_ptr.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), _stream())
        return s, i

    def colbert_topk(self, q_tok: torch.Tensor, k: int):
        nq, q_len = q_tok.shape[0], q_tok.shape[1]
        assert q_tok.is_cuda and q_tok.dtype == torch.float32 and q_tok.is_contiguous()
        s = self._buf("cs", (nq, k), torch.float32)
        i = self._buf("ci", (nq, k), torch.int64)
        self.maxsim.search_device(q_tok.data_ptr(), nq, q_len, k, s.data_ptr(), i.data_ptr(), _stream())
        return s, i

    # -- fusion ---------------------------------------------------------------
    def fuse(self, params: _native.FuseParams, nq: int, dense=None, bm25=None, colbert=None) -> BatchResult:
        def chan(c, m):
            if c is None:
                return None, 0
            s, i = c
            return (i.data_ptr(), s.data_ptr(), int(i.shape[1]), m.data_ptr() if m is not None else 0), int(i.shape[1])
        d, kd = chan(dense, self.maps[0])
        b, kb = chan(bm25, self.maps[1])
        c, kc = chan(colbert, self.maps[2])
        mo = kd + kb + kc
        pk, ids, vals, mask, count = self._fused_outputs(nq, mo)
        _native.fuse_device(params, nq, d, b, c, ids.data_ptr(), vals.data_ptr(), mask.data_ptr(), count.data_ptr(),
                            device=self.device, stream=_stream())
        return BatchResult(ids=ids, vals=vals, mask=mask, count=count, packed=pk)

    def _fused_outputs(self, nq: int, mo: int):
        o1 = nq * mo * 8
        o2 = o1 + nq * mo * _native.FUSE_NVALS * 8
        o3 = o2 + nq * mo * 4
        pk = self._buf("fpk", (o3 + nq * 4,), torch.uint8)  # the four outputs side by side: one D2H serves the host API
        return (pk, pk[:o1].view(torch.int64).view(nq, mo), pk[o1:o2].view(torch.float64).view(nq, mo, _native.FUSE_NVALS),
                pk[o2:o3].view(torch.int32).view(nq, mo), pk[o3:].view(torch.int32))

    def dense_topk_fuse(self, params: _native.FuseParams, q_emb: torch.Tensor, k: int, bm25):
        """Dense top-k + fusion with the finished BM25 lists as ONE native call (amdr_dense_search_fuse_device: for the
        serving corpora under a batch one kernel ranks the score rows and fuses).  Same results as dense_topk + fuse."""
        nq = q_emb.shape[0]
        assert q_emb.is_cuda and q_emb.dtype == torch.float32 and q_emb.is_contiguous()
        bs, bi = bm25
        kb = int(bi.shape[1])
        s = self._buf("ds", (nq, k), torch.float32)
        i = self._buf("di", (nq, k), torch.int64)
        pk, ids, vals, mask, count = self._fused_outputs(nq, k + kb)
        m0, m1 = self.maps[0], self.maps[1]
        self.dense.search_fuse_device(params, q_emb.data_ptr(), nq, k,
                                      (bi.data_ptr(), bs.data_ptr(), kb, m1.data_ptr() if m1 is not None else 0),
                                      m0.data_ptr() if m0 is not None else 0, s.data_ptr(), i.data_ptr(), ids.data_ptr(),
                                      vals.data_ptr(), mask.data_ptr(), count.data_ptr(), _stream())
        return (s, i), BatchResult(ids=ids, vals=vals, mask=mask, count=count, packed=pk)

    def _hybrid_small(self, params: _native.FuseParams, q_emb: torch.Tensor, q_terms: torch.Tensor, q_ptr: torch.Tensor,
                      k: int) -> BatchResult:
        """bm25_topk + dense_topk_fuse through amdr_hybrid_small_device (one launch on a serving corpus).  Issued once
        per query by search(): output tensors and the call's argument block are built once per (nq, k)."""
        nq = int(q_emb.shape[0])
        assert q_emb.is_cuda and q_emb.dtype == torch.float32 and q_emb.is_contiguous()
        assert q_terms.dtype == torch.int32 and q_ptr.dtype == torch.int64 and q_terms.is_cuda and q_ptr.is_cuda
        assert q_ptr.shape[0] - 1 == nq
        ent = self._xcache.get(("hs", nq, k))
        if ent is None:
            ds = self._buf("ds", (nq, k), torch.float32)
            di = self._buf("di", (nq, k), torch.int64)
            bs = self._buf("bs", (nq, k), torch.float64)
            bi = self._buf("bi", (nq, k), torch.int64)
            pk, ids, vals, mask, count = self._fused_outputs(nq, 2 * k)
            m0, m1 = self.maps[0], self.maps[1]
            plan = _native.hybrid_small_plan(self.dense, self.bm25, nq, k, k, m0.data_ptr() if m0 is not None else 0,
                                             m1.data_ptr() if m1 is not None else 0, ds.data_ptr(), di.data_ptr(),
                                             bs.data_ptr(), bi.data_ptr(), ids.data_ptr(), vals.data_ptr(), mask.data_ptr(),
                                             count.data_ptr())
            ent = (plan, (ids, vals, mask, count, pk), (ds, di, bs, bi))
            self._xcache[("hs", nq, k)] = ent
        plan, (ids, vals, mask, count, pk), (ds, di, bs, bi) = ent
        _native.hybrid_small_device(plan, params, q_emb.data_ptr(), q_terms.data_ptr(), q_ptr.data_ptr(), _stream())
        res = BatchResult(ids=ids, vals=vals, mask=mask, count=count, packed=pk)
        res.dense_scores, res.dense_ids = ds, di
        res.bm25_scores, res.bm25_ids = bs, bi
        return res

    def rerank_blend(self, res: BatchResult, ce_raw: torch.Tensor, beta: float) -> BatchResult:
        nq, mo = res.ids.shape
        assert ce_raw.is_cuda and ce_raw.dtype == torch.float64 and ce_raw.is_contiguous() and ce_raw.shape[0] == nq
        out = self._buf("fr", (nq, mo, 2), torch.float64)
        _native.rerank_blend_device(nq, mo, res.count.data_ptr(), res.ids.data_ptr(), res.vals.data_ptr(),
                                    res.mask.data_ptr(), ce_raw.data_ptr(), int(ce_raw.shape[1]), float(beta),
                                    out.data_ptr(), device=self.device, stream=_stream())
        res.rerank = out
        return res

    # -- whole pipeline ---------------------------------------------------------
    def search_batch(self, params: _native.FuseParams, k: int, *, q_emb: Optional[torch.Tensor] = None,
                     q_terms: Optional[torch.Tensor] = None, q_ptr: Optional[torch.Tensor] = None,
                     q_tok: Optional[torch.Tensor] = None) -> BatchResult:
        """dense + bm25 (+ colbert) top-k -> fuse -> min_final filter, all on device."""
        d = b = c = None
        nq = None
        if (self.dense is not None and q_emb is not None and self.bm25 is not None and q_ptr is not None
                and not (self.maxsim is not None and q_tok is not None) and self.shard_offset is None):
            # dense + BM25 on one GPU, the serving hybrid without ColBERT: BM25 first, then the dense channel and the
            # fusion in one native call
            nq = int(q_emb.shape[0])
            if nq <= 4 and 2 * k <= 32:
                # the serving call (search(): one query at a time): both channels and the fusion in ONE launch
                return self._hybrid_small(params, q_emb, q_terms, q_ptr, k)
            b = self.bm25_topk(q_terms, q_ptr, k)
            d, res = self.dense_topk_fuse(params, q_emb, k, b)
            res.dense_scores, res.dense_ids = d
            res.bm25_scores, res.bm25_ids = b
            return res
        with_col = self.maxsim is not None and q_tok is not None
        overlap = (with_col and os.environ.get("AMDR_ENGINE_OVERLAP", "1") != "0"
                   and ((self.dense is not None and q_emb is not None) or (self.bm25 is not None and q_ptr is not None)))
        if overlap:
            # The dense and BM25 channels of a batch are a few short launches that do not fill the chip; MaxSim's first pass is
            # ~0.8 ms on the matrix pipe.  They depend on nothing of each other until the fusion: the two short channels go to
            # a side stream (forked from and joined to the caller's: inside a hipGraph capture the fork / join become edges).
            main = torch.cuda.current_stream(self.tdev)
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=self.tdev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if self.dense is not None and q_emb is not None:
                    d = self.dense_topk(q_emb, k)
                if self.bm25 is not None and q_ptr is not None:
                    b = self.bm25_topk(q_terms, q_ptr, k)
            c = self.colbert_topk(q_tok, k)
            main.wait_stream(side)
            nq = q_tok.shape[0]
        else:
            if self.dense is not None and q_emb is not None:
                d = self.dense_topk(q_emb, k)
                nq = q_emb.shape[0]
            if self.bm25 is not None and q_ptr is not None:
                b = self.bm25_topk(q_terms, q_ptr, k)
                nq = q_ptr.shape[0] - 1
            if with_col:
                c = self.colbert_topk(q_tok, k)
                nq = q_tok.shape[0]
        if self.shard_offset is not None:
            from . import sharding
            chans = [x for x in (d, b, c) if x is not None]
            merged = iter(sharding.exchange_topk(chans, int(self.shard_offset), group=self.shard_group, buf=self._buf,
                                                 cache=self._xcache))
            d, b, c = (next(merged) if x is not None else None for x in (d, b, c))
        res = self.fuse(params, nq, d, b, c)
        if d is not None:
            res.dense_scores, res.dense_ids = d
        if b is not None:
            res.bm25_scores, res.bm25_ids = b
        if c is not None:
            res.colbert_scores, res.colbert_ids = c
        return res

    # -- hipGraph form -----------------------------------------------------------
    def capture(self, params: _native.FuseParams, k: int, *, q_emb: Optional[torch.Tensor] = None,
                q_terms: Optional[torch.Tensor] = None, q_ptr: Optional[torch.Tensor] = None,
                q_tok: Optional[torch.Tensor] = None):
        """Record one search_batch over the given tensors into a hipGraph.

        Returns (graph, result): `graph.replay()` re-runs the whole step — every kernel of every
        stage — as ONE launch on the current stream; new queries are written INTO the same input
        tensors (same nq; the BM25 term array may hold any number of terms up to its length).
        The "_device" entry points only enqueue and, after reserve(), allocate nothing
        (include/amdretrieval.h), which is what makes the step capturable.  A step of 4-5 short
        kernels is launch-bound at small batch: replay removes the per-kernel launch gaps.  (Measured: the BM25
        channel on a forked branch of the captured graph — it does not depend on the dense channel — replays in
        45 us against 33 us for the plain chain: the fork / join nodes cost more than the 5-us kernel they hide.)
        """
        if self.shard_offset is not None:
            raise RuntimeError("capture: a sharded step contains a collective; it is not recorded into a hipGraph")
        nq = (q_emb.shape[0] if q_emb is not None else q_ptr.shape[0] - 1 if q_ptr is not None else q_tok.shape[0])
        self.reserve(int(nq), int(k), int(q_terms.numel()) if q_terms is not None else 0)
        side = torch.cuda.Stream(device=self.tdev)
        side.wait_stream(torch.cuda.current_stream(self.tdev))
        with torch.cuda.stream(side):  # eager warm-up sizes every lazily grown buffer outside the capture
            self.search_batch(params, k, q_emb=q_emb, q_terms=q_terms, q_ptr=q_ptr, q_tok=q_tok)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            res = self.search_batch(params, k, q_emb=q_emb, q_terms=q_terms, q_ptr=q_ptr, q_tok=q_tok)
        return graph, res
