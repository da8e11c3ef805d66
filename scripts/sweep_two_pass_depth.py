import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
import bench
from legal_rag_amd import _native
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for n, d in ((591, 768), (1024, 768)):
    X = torch.randn((n, d), device=dev, generator=g); X /= X.norm(dim=1, keepdim=True)
    idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
    nq = 37376
    Q = torch.randn((nq, d), device=dev, generator=g); Q /= Q.norm(dim=1, keepdim=True)
    st = int(torch.cuda.current_stream().cuda_stream)
    for k in (1, 10, 12, 14, 16, 20, 32):
        s = torch.empty((nq, k), device=dev); i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx.reserve(nq, k)
        r = {}
        for name, flag in (("exact", "0"), ("two", "1")):
            os.environ["AMDR_DENSE_SMALL_HI"] = flag
            r[name] = round(bench.event_ms(torch, lambda: idx.search_device(Q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), st), 10) * 1e3, 1)
        out[f"{n}x{d} k={k}"] = r
    idx.close()
print(json.dumps(out, indent=0))
