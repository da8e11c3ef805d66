import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from legal_rag_amd import _native
W = bench.build_corpus("en", colbert=True)
idx = _native.MaxSimIndex(W["D"], W["doc_ptr"], device=0)
dev = torch.device("cuda", 0)
Q = torch.from_numpy(W["Qtok"]).to(dev)
K = 10
out = {}
for nq in (1, 2, 4, 7, 8, 16, 64):
    q = Q[:nq].contiguous()
    idx.reserve(nq, K)
    s = torch.empty((nq, K), dtype=torch.float32, device=dev); i = torch.empty((nq, K), dtype=torch.int64, device=dev)
    st = int(torch.cuda.current_stream().cuda_stream)
    fn = lambda: idx.search_device(q.data_ptr(), nq, q.shape[1], K, s.data_ptr(), i.data_ptr(), st)
    out[nq] = {"ms": round(bench.event_ms(torch, fn, 50), 4), "plan": idx.plan_info(nq)[:60]}
print(json.dumps(out, indent=0))
