#!/usr/bin/env python3
"""A few launches of the batched dense search on random unit rows, for rocprofv3 (GPU box):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 scripts/run_dense_once.py 591 37376 768 10
Kernel choice follows the library's environment switches (AMDR_DENSE_PANEL, AMDR_PANEL_PARTS)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from legal_rag_amd import _native  # noqa: E402

n, nq, d, reps = (int(x) for x in (sys.argv[1:5] + ["591", "37376", "768", "10"][len(sys.argv) - 1:]))
_native.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
Q = torch.randn((nq, d), device=dev, generator=g)
Q /= Q.norm(dim=1, keepdim=True)
X = torch.randn((n, d), device=dev, generator=g)
X /= X.norm(dim=1, keepdim=True)
idx = _native.DenseIndex(device_ptr=X.data_ptr(), n=n, dim=d, device=0, keepalive=X)
s = torch.empty((nq, 10), dtype=torch.float32, device=dev)
i = torch.empty((nq, 10), dtype=torch.int64, device=dev)
idx.reserve(nq, 10)
st = int(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    idx.search_device(Q.data_ptr(), nq, 10, s.data_ptr(), i.data_ptr(), st)
torch.cuda.synchronize()
idx.close()
