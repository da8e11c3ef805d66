"""Multi-GPU: row-sharded corpus, one process per GPU, RCCL all-gather of the
per-shard top-k over xGMI (the reference is single-process).

Rank r owns the contiguous row block [offset_r, offset_r + n_r) of every
channel, so global id = local id + offset.  Dense, BM25 and MaxSim rows are
independent; BM25's idf and avgdl are corpus-global statistics computed once
and replicated (the shard's CSR keeps only its own documents).  Per query batch
there is exactly ONE collective: every rank packs its per-channel (score bits,
global id) lists into a single int64 buffer and all-gathers it; each rank then
merges W*k -> k per channel (ties -> lower global id) and fuses.  The payload is
B*k*16 bytes per channel per rank — latency-bound on xGMI, never bandwidth-bound
— so the lever is one large collective per batch instead of one per query.

`torch.distributed` is the transport (backend "nccl" == RCCL on ROCm; "gloo" on
CPU for the world_size-2 tests).  The merge itself is the HIP kernel
amdr_merge_topk_*; CPU tests inject a checker merge instead — the product
default has no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .. import _native


@dataclass(frozen=True)
class ShardSpec:
    """This process's place in a row-sharded deployment (one process per GPU)."""
    rank: int
    world: int
    group: Any = None

    def bounds(self, n: int) -> Tuple[int, int]:
        return shard_bounds(int(n), self.world)[self.rank]

    @property
    def key(self) -> Tuple[int, int]:
        return (self.rank, self.world)


def active_shard(rcfg) -> Optional[ShardSpec]:
    """cfg.retrieval.shard = "rows": the retrievers of this process hold the row block of rank r of W in HBM and
    every search exchanges the per-shard top-k (SPMD: every rank makes the same calls with the same arguments and
    gets the same results).  None / "none": unsharded.  The process group is torch.distributed's default group
    (backend nccl = RCCL on a GPU node) unless cfg.retrieval.shard_group names another; asking for shards without
    an initialised process group is an error, never a silent fall-back to a whole-corpus index."""
    mode = getattr(rcfg, "shard", None)
    if mode in (None, "", "none", False):
        return None
    if str(mode) != "rows":
        raise ValueError(f"cfg.retrieval.shard = {mode!r}: only 'rows' (contiguous row blocks) is implemented")
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("cfg.retrieval.shard = 'rows' needs an initialised torch.distributed process group "
                           "(one process per GPU)")
    group = getattr(rcfg, "shard_group", None)
    world = dist.get_world_size(group)
    if world == 1:
        return None
    return ShardSpec(rank=dist.get_rank(group), world=world, group=group)


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks [r*n/W, (r+1)*n/W)."""
    return [((r * n) // world, ((r + 1) * n) // world) for r in range(world)]


def to_global(ids: torch.Tensor, offset: int) -> torch.Tensor:
    return torch.where(ids >= 0, ids + offset, ids)


def pack_channels(chans: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """[(scores[nq,k_c] f32|f64, global ids[nq,k_c] i64)] -> int64 [nq, sum_c 2*k_c]."""
    parts = []
    for s, i in chans:
        bits = s.to(torch.float64).view(torch.int64)  # f32 -> f64 widening is exact
        parts += [bits, i]
    return torch.cat(parts, dim=1).contiguous()


def unpack_channels(buf: torch.Tensor, ks: Sequence[int], dtypes: Sequence[torch.dtype]):
    """inverse of pack_channels on a gathered [W, nq, sum 2k] buffer."""
    out = []
    col = 0
    for k, dt in zip(ks, dtypes):
        s = buf[..., col:col + k].contiguous().view(torch.float64).to(dt).contiguous()
        i = buf[..., col + k:col + 2 * k].contiguous()
        out.append((s, i))
        col += 2 * k
    return out


def native_merge(scores: torch.Tensor, ids: torch.Tensor, k_out: int):
    """scores/ids [W, nq, k] on the GPU -> merged [nq, k_out] (HIP kernel)."""
    if not scores.is_cuda:
        raise RuntimeError("native_merge needs CUDA tensors: there is no CPU merge in the product path")
    W, nq, k = scores.shape
    f64 = scores.dtype == torch.float64
    out_s = torch.empty((nq, k_out), dtype=scores.dtype, device=scores.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    _native.merge_topk_device(scores.data_ptr(), ids.data_ptr(), W, nq, k, k_out, out_s.data_ptr(), out_i.data_ptr(),
                              f64=f64, device=scores.device.index or 0,
                              stream=int(torch.cuda.current_stream().cuda_stream))
    return out_s, out_i


def exchange_topk_native(chans: Sequence[Tuple[torch.Tensor, torch.Tensor]], offset: int, *, group=None,
                         buf: Optional[Callable] = None, cache: Optional[dict] = None):
    """The product form of the exchange for device-resident lists: ONE pack launch (amdr_shard_pack_device: every
    channel's score bits + global ids straight into the all-gather's send buffer), the all-gather, ONE merge launch
    (amdr_shard_merge_device: reads the gathered buffer in place, all channels).  `buf(name, shape, dtype)` supplies
    persistent buffers (HybridEngine._buf: no allocation per call); `cache` (a dict the caller keeps) remembers the
    argument blocks of the two native calls per set of input tensors, so a steady-state call is two ctypes calls
    and the collective."""
    dev = chans[0][0].device
    nq = int(chans[0][0].shape[0])
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    key = (tuple((s.data_ptr(), i.data_ptr(), tuple(s.shape), s.dtype) for s, i in chans), int(offset), world)
    plan = cache.get(key) if cache is not None else None
    if plan is None:
        for s, i in chans:
            if not (s.is_cuda and i.is_cuda and s.is_contiguous() and i.is_contiguous() and i.dtype == torch.int64
                    and s.dtype in (torch.float32, torch.float64) and s.shape == i.shape and s.shape[0] == nq):
                raise ValueError("exchange_topk_native: channels must be contiguous CUDA (scores f32|f64, ids i64) [nq, k]")
        mk = buf or (lambda name, shape, dtype: torch.empty(shape, dtype=dtype, device=dev))
        ks = [int(i.shape[1]) for _, i in chans]
        row = _native.shard_row_words(ks)
        send = mk("xsend", (nq, row), torch.int64)
        gathered = send if world == 1 else mk("xrecv", (world * nq, row), torch.int64)
        out = [(mk(f"xs{c}", (nq, k), s.dtype), mk(f"xi{c}", (nq, k), torch.int64))
               for c, ((s, _), k) in enumerate(zip(chans, ks))]
        pack_args = _native.shard_chans([(s.data_ptr(), i.data_ptr(), k, s.dtype == torch.float64)
                                         for (s, i), k in zip(chans, ks)])
        merge_args = _native.shard_chans([(os_.data_ptr(), oi.data_ptr(), k, os_.dtype == torch.float64)
                                          for (os_, oi), k in zip(out, ks)])
        plan = (pack_args, merge_args, send, gathered, out, list(chans))  # (the inputs kept alive with their pointers)
        if cache is not None:
            if len(cache) > 64:
                cache.clear()
            cache[key] = plan
    pack_args, merge_args, send, gathered, out, _ = plan
    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    di = dev.index or 0
    _native.shard_pack_args(pack_args, nq, int(offset), send.data_ptr(), device=di, stream=stream)
    if world > 1:
        dist.all_gather_into_tensor(gathered, send, group=group)
    _native.shard_merge_args(gathered.data_ptr(), world, nq, merge_args, device=di, stream=stream)
    return out


def exchange_topk(chans: Sequence[Tuple[torch.Tensor, torch.Tensor]], offset: int, *, group=None,
                  merge_fn: Optional[Callable] = None, buf: Optional[Callable] = None, cache: Optional[dict] = None):
    """All-gather the local per-channel top-k of this rank's shard and merge.

    chans: [(scores[nq,k_c], LOCAL ids[nq,k_c])].  Returns the same structure
    holding the global top-k (global ids), identical on every rank.  Device-resident lists take the two native
    launches (exchange_topk_native); the torch form below is what the CPU tests drive with an injected checker
    merge (the product has no CPU merge)."""
    if merge_fn is None and chans and all(s.is_cuda and i.is_cuda for s, i in chans) and len(chans) <= 4:
        return exchange_topk_native([(s.contiguous(), i.contiguous()) for s, i in chans], offset, group=group, buf=buf,
                                    cache=cache)
    merge_fn = merge_fn or native_merge
    ks = [int(i.shape[1]) for _, i in chans]
    dts = [s.dtype for s, _ in chans]
    local = pack_channels([(s, to_global(i, offset)) for s, i in chans])
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        gathered = local.unsqueeze(0)
    else:
        # concatenated-along-dim-0 output form: accepted by both RCCL and gloo
        flat = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, local, group=group)
        gathered = flat.view(world, local.shape[0], local.shape[1])
    out = []
    for (s, i), k in zip(unpack_channels(gathered, ks, dts), ks):
        out.append(merge_fn(s, i, k))
    return out


def exchange_topk_numpy(chans: Sequence[Tuple[np.ndarray, np.ndarray]], offset: int, device: int, *, group=None):
    """Host-array form for the per-channel API (VectorStore.search / BM25Retriever.search / ColBERTRetriever.search
    in shard mode): local (scores, LOCAL ids) numpy arrays up to the card, exchange_topk there (the merge is the
    HIP kernel on every backend), merged global lists back as numpy."""
    tdev = torch.device("cuda", int(device))
    up = [(torch.from_numpy(np.ascontiguousarray(s)).to(tdev), torch.from_numpy(np.ascontiguousarray(i)).to(tdev))
          for s, i in chans]
    with torch.cuda.device(tdev):
        out = exchange_topk(up, offset, group=group)
    return [(s.cpu().numpy(), i.cpu().numpy()) for s, i in out]


def allreduce_max_numpy(a: np.ndarray, device: int, *, group=None) -> np.ndarray:
    """Element-wise maximum over the ranks (graph rescoring: every candidate row lives on exactly one shard, the
    others contribute -inf)."""
    t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", int(device)))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy()
