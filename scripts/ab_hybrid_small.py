"""A/B of the serving call (dense + BM25 + fusion of ONE query on a serving corpus): the separate launches
(AMDR_HYBRID_SMALL=0) against the one-launch step (csrc/fuse.hip hybrid_small_kernel).  Per corpus shape: eager p50 / p90
of HybridEngine.search_batch at 1, 2 and 4 queries (host call + device time, synchronised per call) and the same step
replayed from a hipGraph.  AMDR_HYBRID_SMALL_ROWS (chunk rows per dense block) is read once per process."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from legal_rag_amd import _native  # noqa: E402
from legal_rag_amd.retrieval.engine import HybridEngine  # noqa: E402
from oracle import bm25 as OB  # noqa: E402  (corpus builder only: the timed path is the native one)


def corpus(rng, n, d):
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    words = [f"w{i}" for i in range(3000)]
    docs = [[words[j] for j in rng.integers(0, 3000, size=int(rng.integers(30, 120)))] for _ in range(n)]
    ob = OB.BM25Okapi(docs)
    csr = OB.to_csr(ob)
    eng = HybridEngine(_native.DenseIndex(X), _native.BM25Index(csr["term_ptr"], csr["post_doc"], csr["post_tf"], csr["idf"],
                                                                csr["doc_len"], ob.avgdl, ob.k1, ob.b), None)
    return eng, len(csr["vocab"])


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return round(ts[len(ts) // 2], 1), round(ts[int(len(ts) * 0.9)], 1)


def main():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    k = 10
    params = _native.make_fuse_params()
    # what an eager call cannot go under on this box: one trivial launch + the same synchronisation
    x = torch.zeros(64, device=dev)
    for _ in range(50):
        x.fill_(1.0)
    torch.cuda.synchronize()
    f50, f90 = timed(lambda: x.fill_(1.0), 400)
    st = torch.cuda.current_stream()
    s50, _ = timed(lambda: (x.fill_(1.0), st.synchronize()), 400)
    print(json.dumps({"floor_fill_plus_device_sync_p50_us": f50, "p90": f90, "fill_plus_stream_sync_p50_us": s50}), flush=True)
    shapes = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("AB_SHAPES", "591:384,1260:768,2048:768").split(",")]
    for n, d in shapes:
        eng, V = corpus(rng, n, d)
        for nq in (1, 2, 4):
            q = rng.standard_normal((nq, d)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            Q = torch.from_numpy(q).to(dev)
            qt_h, qp_h = _native.BM25Index.pack_queries([[int(t) for t in rng.integers(0, V, size=19)] for _ in range(nq)])
            qt, qp = torch.from_numpy(qt_h).to(dev), torch.from_numpy(qp_h).to(dev)
            eng.reserve(nq, k, int(qp_h[-1]))
            out = {"n": n, "d": d, "nq": nq, "rows_per_block": os.environ.get("AMDR_HYBRID_SMALL_ROWS", "default")}
            for name, flag in (("separate", "0"), ("one_launch", "1")):
                os.environ["AMDR_HYBRID_SMALL"] = flag
                step = lambda: eng.search_batch(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)  # noqa: E731
                for _ in range(50):
                    step()
                torch.cuda.synchronize()
                p50, p90 = timed(step, 400)
                t0 = time.perf_counter()
                for _ in range(200):
                    step()
                enq = (time.perf_counter() - t0) / 200 * 1e6  # host cost of the call alone (the queue absorbs the kernels)
                torch.cuda.synchronize()
                g, _res = eng.capture(params, k, q_emb=Q, q_terms=qt, q_ptr=qp)
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                g50, g90 = timed(g.replay, 400)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(200):
                    g.replay()
                b.record()
                torch.cuda.synchronize()
                out[name] = {"enqueue_us": round(enq, 1), "eager_p50_us": p50, "eager_p90_us": p90, "graph_p50_us": g50, "graph_p90_us": g90,
                             "graph_device_us": round(a.elapsed_time(b) * 1e3 / 200, 2)}
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
