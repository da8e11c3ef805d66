"""Time of the shard exchange's two launches (pack + merge) at world = 1 — the per-rank fixed cost of a sharded step
that is measurable on one GPU (<= 25 us for 2 376 queries x 3 channels x top-10) — beside the
torch form of rounds 1-3 (where / to / cat / contiguous per channel + one merge launch per channel)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from legal_rag_amd.retrieval import sharding  # noqa: E402


def timed(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for nq, k in ((2376, 10), (64, 10), (256, 10), (2376, 80)):
        g = torch.Generator().manual_seed(1)
        chans = []
        for dt in (torch.float32, torch.float64, torch.float32):
            s = torch.sort(torch.randn((nq, k), generator=g, dtype=dt), dim=1, descending=True).values.to(dev)
            i = torch.stack([torch.randperm(1000, generator=g)[:k] for _ in range(nq)]).to(dev)
            chans.append((s, i))
        bufs = {}

        def buf(name, shape, dtype):
            key = (name, tuple(shape), dtype)
            if key not in bufs:
                bufs[key] = torch.empty(shape, dtype=dtype, device=dev)
            return bufs[key]
        xc = {}
        native = timed(lambda: sharding.exchange_topk(chans, 1000, buf=buf, cache=xc))
        torch_form = timed(lambda: sharding.exchange_topk(chans, 1000, merge_fn=sharding.native_merge), iters=50, warm=5)
        out[f"{nq}x3x{k}"] = {"native_pack_plus_merge_us": round(native, 2), "torch_form_us": round(torch_form, 2)}
    print(json.dumps({"exchange_topk_world1": out}))


if __name__ == "__main__":
    main()
