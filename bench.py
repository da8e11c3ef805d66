#!/usr/bin/env python3
"""Benchmark of the hybrid-retrieval hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ucc_hybrid|synth10m]

Contract (driver): one JSON line on rank 0.  metric = queries/sec (+ Recall@10
of the hybrid top-10) on the UCC-en corpus, BASELINE.json configs[1]
(dense + BM25 + fusion, top-10, 1 x MI355X).  A "step" = one pass of the hot
path (dense scan + top-k, BM25 scoring + top-k, fusion, min_final filter) over
one batch of queries whose embeddings / term ids are already resident in HBM.
N > 1: one process per GPU; for the UCC-en workload every rank holds a replica
of the 1.8 MB corpus and answers its own batch ("weak", no data-path
collective); `--shard corpus` row-shards the corpus instead and adds the RCCL
all-gather + merge of the per-shard top-k (the layout used when the chunk
matrix does not fit one GPU, e.g. --workload synth10m).

Extra objects in the same line:
  roofline      dominant kernel (dense scan) of the timed region, HIP events
                bracketing that kernel on its launch stream
  cpu_baseline  the oracle (oracle/, numpy) timed on this box's host cores
  hbm_scan      the dense channel on the synthetic 10M x 768 matrix (BASELINE.json
                configs[4]) at 4 and 32 queries per scan — the HBM-roofline evidence
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA == fp32 vector peak


def pmc_traffic(key):
    """HBM bytes per launch measured with rocprofv3 PMC in a separate run of this same
    command (profiles/pmc_traffic.json); None if no measurement is on record."""
    try:
        rec = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get(key)
        return float(rec["bytes_per_launch"]) if rec else None
    except Exception:  # noqa: BLE001
        return None


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ucc_hybrid", choices=["ucc_hybrid", "ucc_colbert", "synth10m"])
    ap.add_argument("--shard", default="auto", choices=["auto", "queries", "corpus"])
    ap.add_argument("--repeat", type=int, default=32, help="ucc_hybrid: the query set is tiled this many times per step")
    ap.add_argument("--synth-rows", type=int, default=10_000_000)
    ap.add_argument("--synth-batch", type=int, default=8)
    ap.add_argument("--no-hbm-scan", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------
def build_ucc(device: int, colbert: bool = False):
    """UCC-en corpus -> dense matrix (stand-in BGE embeddings), BM25 index, query set."""
    from legal_rag_amd import _native, text
    from legal_rag_amd.bm25_model import BM25Okapi
    from legal_rag_amd.encoders import HashingEmbedder
    from legal_rag_amd.evaluation import synthetic_queries
    from legal_rag_amd.retrieval.corpus_loader import load_chunks_from_dir

    chunks = load_chunks_from_dir(str(ROOT / "tests" / "golden" / "corpus"), "law_en.jsonl")
    emb = HashingEmbedder(dim=768)
    X = emb.encode([c.text for c in chunks])
    corpus_tokens = [text.tokenize_en(c.text) for c in chunks]
    bm = BM25Okapi(corpus_tokens)
    qs = synthetic_queries(chunks, seed=0)
    Q = emb.encode_queries([q for q, _, _ in qs])
    q_tokens = [text.jieba_cut(q) for q, _, _ in qs]   # bm25_retriever.py:73 (not lower-cased)
    q_tid = [bm.term_ids(t) for t in q_tokens]
    W = dict(chunks=chunks, X=X, bm=bm, corpus_tokens=corpus_tokens, queries=qs, Q=Q, q_tokens=q_tokens, q_tid=q_tid)
    if colbert:
        from legal_rag_amd.encoders import HashingTokenEmbedder
        te = HashingTokenEmbedder()
        mats = [te.encode_doc(c.text.strip()) for c in chunks]
        W["D"] = np.concatenate(mats, axis=0)
        W["doc_ptr"] = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
        W["Qtok"] = np.stack([te.encode_query(q.strip()) for q, _, _ in qs]).astype(np.float32)
    return W


def hybrid_recall(ids_top, counts, chunks, queries, k=10):
    art = [c.article_id for c in chunks]
    tot = 0.0
    for qi, (_, gold, _) in enumerate(queries):
        n = min(int(counts[qi]), k)
        pred = list(dict.fromkeys(art[int(i)] for i in ids_top[qi, :n] if i >= 0))
        tot += 1.0 if gold in pred[:k] else 0.0
    return tot / max(len(queries), 1)


def oracle_pipeline(W, qi_list, k=10):
    """CPU oracle of the same step for the listed queries -> list of id lists."""
    from oracle import bm25 as OB
    from oracle import dense as OD
    from oracle import fusion as OF
    ob = W["_oracle_bm25"]
    S, I = OD.flatip_topk(W["X"], W["Q"][qi_list], k)
    out = []
    for j, qi in enumerate(qi_list):
        d = [(int(i), float(s)) for s, i in zip(S[j], I[j]) if i >= 0]
        b = OB.search(ob, W["q_tokens"][qi], k)
        c = []
        if "D" in W:
            from oracle import maxsim as OM
            cs, ci = OM.maxsim_topk(W["Qtok"][qi][None], W["D"], W["doc_ptr"], k)
            c = [(int(i), float(np.float32(s))) for s, i in zip(cs[0], ci[0]) if i >= 0]
        fused = OF.fuse(d, b, c, {})
        fused = [h for h in fused if h["score"] >= 0.2]
        out.append([h["id"] for h in fused[:k]])
    return out


def cpu_baseline(W, seconds: float):
    from oracle import bm25 as OB
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        blas_threads = os.cpu_count() or 1
    W["_oracle_bm25"] = OB.BM25Okapi(W["corpus_tokens"])
    nq = len(W["queries"])
    done = 0
    t0 = time.perf_counter()
    chunk = 64
    pos = 0
    while True:
        idx = [(pos + j) % nq for j in range(chunk)]
        oracle_pipeline(W, idx)
        done += chunk
        pos = (pos + chunk) % nq
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": done / el, "unit": "queries/s", "cores": int(blas_threads), "kind": "port",
            "sample": f"{done} UCC-en hybrid queries (numpy fp32 X@Q.T batches of {chunk} + exact top-10, "
                      f"rank_bm25-restated fp64 get_scores + stable sort, python fusion) in {el:.1f}s; "
                      f"BLAS threads={blas_threads}, BM25/fusion single-threaded python; host has {os.cpu_count()} cpus"}


# ---------------------------------------------------------------------------
def synth_matrix(torch, n, d, device, seed=1234):
    """rng(seed) normal rows, L2-normalised, generated in HBM in 1M-row chunks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1_000_000
    for s in range(0, n, step):
        e = min(n, s + step)
        blk = torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        blk /= blk.norm(dim=1, keepdim=True)
        X[s:e] = blk
    return X


def run_hbm_scan(torch, device, n, d, batches, steps, warmup, k=10):
    from legal_rag_amd import _native
    X = synth_matrix(torch, n, d, device)
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    Q = torch.randn((1024, d), generator=g, device=device, dtype=torch.float32)
    Q /= Q.norm(dim=1, keepdim=True)
    out = []
    for B in batches:
        out.append(_hbm_scan_one(torch, _native, device, X, Q, n, d, B, steps, warmup, k))
    del X
    torch.cuda.empty_cache()
    return out


def _hbm_scan_one(torch, _native, device, X, Q, n, d, B, steps, warmup, k):
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=device.index, keepalive=X)
    idx.reserve(B, k)
    s = torch.empty((B, k), dtype=torch.float32, device=device)
    i = torch.empty((B, k), dtype=torch.int64, device=device)
    st = int(torch.cuda.current_stream().cuda_stream)
    for w in range(warmup):
        idx.search_device(Q[(w * B) % 1024:].data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
    torch.cuda.synchronize()
    idx.profile_begin(steps)
    t0 = time.perf_counter()
    for w in range(steps):
        off = (w * B) % (1024 - B + 1)
        idx.search_device(Q[off:].data_ptr(), B, k, s.data_ptr(), i.data_ptr(), st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    scan_ms, launches = idx.profile_end()
    # parity on a prefix: oracle over the first 200k rows must agree with a scan of that prefix
    from oracle import dense as OD
    npre = min(n, 200_000)
    pre = _native.DenseIndex(device_ptr=X.data_ptr(), n=npre, dim=d, device=device.index, keepalive=X)
    Qh = Q[:B].cpu().numpy()
    gs, gi = pre.search(Qh, k)
    es, ei = OD.flatip_topk(X[:npre].cpu().numpy(), Qh, k)
    agree = float(np.mean(gi == ei))
    maxerr = float(np.max(np.abs(gs - es)))
    batched = B >= 5  # 32-query-tile fp32-MFMA form: scores S[B, n] are written once and read once
    bytes_per_launch = float(n) * d * 4 + B * d * 4 + (float(n) * B * 4 if batched else B * k * 8)
    per_launch_ms = scan_ms / max(launches, 1)
    achieved = bytes_per_launch / (per_launch_ms * 1e-3) / 1e9
    out = {"workload": f"synthetic {n}x{d} fp32 rows in HBM, {B} queries/scan, top-{k}",
           "kernel": "dense_mfma_scores_kernel" if batched else "dense_scan_topk_kernel",
           "queries_per_s": B * steps / wall, "ms_per_scan_wall": wall / steps * 1e3,
           "scan_kernel_ms": per_launch_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
           "achieved_GBs": achieved, "peak_GBs": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
           "traffic": pmc_traffic(f"synth10m_b{B}/" + ("dense_mfma_scores_kernel" if batched else "dense_scan_topk_kernel"))
           if (n == 10_000_000 and d == 768) else None,
           "f32_TFLOPs": 2.0 * n * d * B / (per_launch_ms * 1e-3) / 1e12,
           "oracle_prefix_rows": npre, "oracle_id_agreement": agree, "oracle_max_abs_err": maxerr}
    idx.close()
    pre.close()
    return out


def with_colbert_tokens(result, scope):
    """Annotate the colbert workload with the MaxSim work per step (no timing here)."""
    W = scope.get("W")
    if isinstance(W, dict) and "doc_ptr" in W and isinstance(result, dict) and "config" in result:
        tokens = int(W["doc_ptr"][-1])
        result["config"]["colbert_doc_tokens"] = tokens
        result["config"]["maxsim_gflop_per_query"] = 2.0 * 32 * 128 * tokens / 1e9
    return False


# ---------------------------------------------------------------------------
def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (single-GPU box): BENCH_FORCE_DEVICE=0 puts every rank on one card,
    # BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if a.gpus != world:
        log(f"note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from legal_rag_amd import _native
    from legal_rag_amd.retrieval import sharding
    from legal_rag_amd.retrieval.engine import HybridEngine

    _native.load()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = {}
    if a.workload in ("ucc_hybrid", "ucc_colbert"):
        K = 10
        with_colbert = a.workload == "ucc_colbert"
        W = build_ucc(local, colbert=with_colbert)
        nq0 = len(W["queries"])
        rep = 1 if with_colbert else max(1, a.repeat)
        shard = "queries" if a.shard == "auto" else a.shard
        n = W["X"].shape[0]
        if shard == "corpus" and world > 1:
            lo, hi = sharding.shard_bounds(n, world)[rank]
        else:
            lo, hi = 0, n
        # ---- index in HBM (this rank's rows) -------------------------------
        dense = _native.DenseIndex(W["X"][lo:hi], device=local)
        bm = W["bm"]
        if (lo, hi) != (0, n):
            # doc-partitioned postings, GLOBAL idf / avgdl
            tp, pd, pt, idf, dl = bm.to_csr()
            keep = (pd >= lo) & (pd < hi)
            cnt = np.zeros(len(tp), dtype=np.int64)
            term_of = np.repeat(np.arange(len(tp) - 1), np.diff(tp))
            np.add.at(cnt, term_of[keep] + 1, 1)
            bmi = _native.BM25Index(np.cumsum(cnt), pd[keep] - lo, pt[keep], idf, dl[lo:hi], float(bm.avgdl),
                                    bm.k1, bm.b, device=local)
        else:
            bmi = bm.gpu(local)
        msi = None
        q_tok = None
        if with_colbert:
            if (lo, hi) != (0, n):
                raise SystemExit("ucc_colbert: use --shard queries")
            msi = _native.MaxSimIndex(W["D"], W["doc_ptr"], device=local)
            q_tok = torch.from_numpy(np.tile(W["Qtok"], (rep, 1, 1))).to(device)
        eng = HybridEngine(dense, bmi, msi, device=local)
        # ---- this rank's query batch, resident in HBM ----------------------
        Qh = np.tile(W["Q"], (rep, 1))
        tids = W["q_tid"] * rep
        q_terms_h, q_ptr_h = _native.BM25Index.pack_queries(tids)
        q_emb = torch.from_numpy(Qh).to(device)
        q_terms = torch.from_numpy(q_terms_h).to(device)
        q_ptr = torch.from_numpy(q_ptr_h).to(device)
        nq = q_emb.shape[0]
        params = _native.make_fuse_params(min_final_score=0.2)  # reference defaults (config.py:92-94,128)
        eng.reserve(nq, K, int(q_ptr_h[-1]))

        def step():
            if shard == "corpus" and world > 1:
                d = eng.dense_topk(q_emb, K)
                b = eng.bm25_topk(q_terms, q_ptr, K)
                (ds, di), (bs, bi) = sharding.exchange_topk([d, b], lo)
                return eng.fuse(params, nq, (ds, di), (bs, bi), None)
            return eng.search_batch(params, K, q_emb=q_emb, q_terms=q_terms, q_ptr=q_ptr, q_tok=q_tok)

        for _ in range(a.warmup):
            step()
        barrier()
        dense.profile_begin(a.steps)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res = step()
        barrier()
        dt = time.perf_counter() - t0
        scan_ms, launches = dense.profile_end()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        queries_per_step_total = nq * (world if shard == "queries" else 1)
        value = queries_per_step_total * a.steps / dt

        ids = res.ids[:, :K].cpu().numpy()
        cnt = res.count.cpu().numpy()
        rec = hybrid_recall(ids[:nq0], cnt[:nq0], W["chunks"], W["queries"], K)
        # ---- single-query latency through the same kernels (B = 1) ----------
        lat_us = lat_p50 = lat_p99 = lat_graph_p50 = lat_graph_p99 = None
        if rank == 0:
            q1 = q_emb[:1].contiguous()
            t1 = q_terms[: int(q_ptr_h[1])].contiguous() if q_ptr_h[1] > 0 else q_terms[:1]
            p1 = q_ptr[:2].contiguous()
            for _ in range(20):
                eng.search_batch(params, K, q_emb=q1, q_terms=t1, q_ptr=p1)
            torch.cuda.synchronize()
            lat = []
            for _ in range(300):
                tt = time.perf_counter()
                eng.search_batch(params, K, q_emb=q1, q_terms=t1, q_ptr=p1)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - tt) * 1e6)
            lat.sort()
            lat_us = sum(lat) / len(lat)
            lat_p50, lat_p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.99)]
            # the same single-query step recorded once into a hipGraph and replayed (one launch per query)
            graph, _gres = eng.capture(params, K, q_emb=q1, q_terms=t1, q_ptr=p1)
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            glat = []
            for _ in range(300):
                tt = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                glat.append((time.perf_counter() - tt) * 1e6)
            glat.sort()
            lat_graph_p50, lat_graph_p99 = glat[len(glat) // 2], glat[int(len(glat) * 0.99)]

        rows_local = hi - lo
        d = W["X"].shape[1]
        # Dominant kernel of the step = dense_mfma_scores_kernel: [rows x d] . [d x nq] in exact fp32 on
        # the matrix pipe.  At UCC-en size X (1.8 MB) is L2-resident, so the bound is the fp32 MFMA rate
        # (157.3 TFLOP/s = the fp32 vector rate, MI355X_MICROARCH.md), not HBM.
        flops_per_launch = 2.0 * rows_local * d * nq
        per_launch_ms = scan_ms / max(launches, 1)
        achieved = flops_per_launch / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "dense_mfma_scores_kernel (v_mfma_f32_16x16x4_f32, exact fp32)",
                    "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                    "traffic": pmc_traffic("ucc_hybrid/dense_mfma_scores_kernel") if (not with_colbert and rep == 32
                                                                                     and shard == "queries") else None,
                    "launch_ms": per_launch_ms,
                    "algorithmic_flops": flops_per_launch,
                    "algorithmic_bytes": float(rows_local) * d * 4 + nq * d * 4 + float(rows_local) * nq * 4,
                    "note": "HBM-roofline evidence for the same channel on a 30.7 GB matrix is in hbm_scan"}
        if with_colbert:
            # the MaxSim pass dominates this workload: time it on its own (torch events see the stream it runs on)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, min(a.steps, 10))
            eng.colbert_topk(q_tok, K)
            e0.record()
            for _ in range(reps):
                eng.colbert_topk(q_tok, K)
            e1.record()
            torch.cuda.synchronize()
            ms_ms = e0.elapsed_time(e1) / reps
            tokens = int(W["doc_ptr"][-1])
            ms_flops = 2.0 * q_tok.shape[1] * q_tok.shape[2] * tokens * nq
            roofline = {"bound": "mfma", "kernel": "maxsim_scores_blocked_kernel + rowscores_topk_kernel (v_mfma_f32_16x16x4_f32)",
                        "achieved": ms_flops / (ms_ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ms_flops / (ms_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "launch_ms": ms_ms, "algorithmic_flops": ms_flops,
                        "algorithmic_bytes": float(tokens) * q_tok.shape[2] * 4 + float(nq) * q_tok.shape[1] * q_tok.shape[2] * 4,
                        "note": "MaxSim channel (scores + top-k launches together); the dense channel's kernel is "
                                "reported by the default workload"}
        result = {
            "metric": "queries/sec + Recall@10 (hybrid top-10) on UCC-en", "value": value, "unit": "queries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if shard == "queries" else "strong", "vs_baseline": None,
            "dtype": "f32 dense / f64 bm25+fusion", "data": "UCC-en law text (fixture) with deterministic stand-in "
            "embeddings (no BGE weights offline) and a seeded synthetic query set",
            "config": {"workload": ("UCC-en 591 chunks, dense(768-d FlatIP)+BM25+ColBERT MaxSim hybrid fusion top-10 "
                                    "(BASELINE configs[2])" if with_colbert else
                                    "UCC-en 591 chunks, dense(768-d FlatIP)+BM25 hybrid fusion top-10 "
                                    "(BASELINE configs[1])"), "queries_per_step_per_gpu": nq, "unique_queries": nq0,
                       "top_k": K, "shard": shard, "fusion": "rrf_norm_blend w=0.6/0.4 alpha=0.5 rrf_k=60 "
                       "min_final=0.2"},
            "recall_at_10": rec, "latency_b1_us": lat_us, "latency_b1_p50_us": lat_p50, "latency_b1_p99_us": lat_p99,
            "latency_b1_graph_p50_us": lat_graph_p50, "latency_b1_graph_p99_us": lat_graph_p99,
            "roofline": roofline,
        }
        if rank == 0:
            # agreement@10 with the CPU oracle on identical inputs
            from oracle import bm25 as OB
            W["_oracle_bm25"] = OB.BM25Okapi(W["corpus_tokens"])
            sample = list(range(0, nq0, max(1, nq0 // (32 if with_colbert else 256))))
            exp = oracle_pipeline(W, sample, K)
            same = 0
            for j, qi in enumerate(sample):
                got = [int(x) for x in ids[qi, :min(int(cnt[qi]), K)]]
                same += int(got == exp[j])
            result["agreement_at_10_vs_oracle"] = same / len(sample)
            if not with_colbert and shard == "queries":
                # the evaluation depth of the reference (evaluate_retrieval.py: k = 80), one untimed pass
                K80 = 80
                s80 = list(range(0, nq0, max(1, nq0 // 64)))
                r80 = eng.search_batch(params, K80, q_emb=q_emb[:nq0].contiguous(), q_terms=q_terms[: int(q_ptr_h[nq0])],
                                       q_ptr=q_ptr[: nq0 + 1].contiguous())
                ids80, cnt80 = r80.ids.cpu().numpy(), r80.count.cpu().numpy()
                exp80 = oracle_pipeline(W, s80, K80)
                ok80 = sum(int([int(x) for x in ids80[qi, :min(int(cnt80[qi]), K80)]] == exp80[j])
                           for j, qi in enumerate(s80))
                result["agreement_at_80_vs_oracle"] = ok80 / len(s80)
                result["recall_at_80"] = hybrid_recall(ids80, cnt80, W["chunks"], W["queries"], K80)
            if not a.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(W, a.cpu_seconds)
        dense.close()
        del eng
    else:
        K = 10
        n_total = a.synth_rows
        B = a.synth_batch
        # corpus row-sharded over ranks (strong scaling of one 10M-row corpus)
        lo, hi = sharding.shard_bounds(n_total, world)[rank]
        X = synth_matrix(torch, hi - lo, 768, device, seed=1234 + rank)
        g = torch.Generator(device=device)
        g.manual_seed(4321)
        Q = torch.randn((1024, 768), generator=g, device=device, dtype=torch.float32)
        Q /= Q.norm(dim=1, keepdim=True)
        idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=hi - lo, dim=768, device=local, keepalive=X)
        idx.reserve(B, K)
        s = torch.empty((B, K), dtype=torch.float32, device=device)
        i = torch.empty((B, K), dtype=torch.int64, device=device)

        def step(w):
            off = (w * B) % (1024 - B + 1)
            idx.search_device(Q[off:].data_ptr(), B, K, s.data_ptr(), i.data_ptr(),
                              int(torch.cuda.current_stream().cuda_stream))
            if world > 1:
                return sharding.exchange_topk([(s, i)], lo)
            return [(s, i)]

        for w in range(a.warmup):
            step(w)
        barrier()
        idx.profile_begin(a.steps)
        t0 = time.perf_counter()
        for w in range(a.steps):
            step(w)
        barrier()
        dt = time.perf_counter() - t0
        scan_ms, launches = idx.profile_end()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        batched = B >= 5  # 32-query-tile fp32-MFMA form: S[B, rows] is written once (and read once by the top-k pass)
        bytes_per_launch = float(hi - lo) * 768 * 4 + B * 768 * 4 + (float(hi - lo) * B * 4 if batched else B * K * 8)
        per_launch_ms = scan_ms / max(launches, 1)
        achieved = bytes_per_launch / (per_launch_ms * 1e-3) / 1e9
        result = {
            "metric": "queries/sec, brute-force cosine top-10", "value": B * a.steps / dt, "unit": "queries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": f"synthetic {n_total}x768 fp32 chunk matrix row-sharded over "
                                            f"{world} GPU(s), {B} queries/scan, top-10 (BASELINE configs[4])"},
            "roofline": {"bound": "hbm", "kernel": "dense_mfma_scores_kernel" if batched else "dense_scan_topk_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(f"synth10m_b{B}/" + ("dense_mfma_scores_kernel" if batched else
                                                                        "dense_scan_topk_kernel"))
                         if (world == 1 and n_total == 10_000_000) else None,
                         "launch_ms": per_launch_ms, "algorithmic_bytes": bytes_per_launch,
                         "f32_TFLOPs": 2.0 * (hi - lo) * 768 * B / (per_launch_ms * 1e-3) / 1e12},
        }
        idx.close()
        del X

    if with_colbert_tokens(result, locals()):
        pass
    if rank == 0 and a.workload == "ucc_hybrid" and not a.no_hbm_scan:
        torch.cuda.empty_cache()
        try:
            result["hbm_scan"] = run_hbm_scan(torch, device, a.synth_rows, 768, [4, 32], steps=20, warmup=3)
        except Exception as e:  # noqa: BLE001 - report, never hide
            result["hbm_scan"] = {"error": repr(e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
