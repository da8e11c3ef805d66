"""Calibration for a two-pass / split-fp16 form of the long-batch dense step (37 376 queries x 591 chunks x 768): what
library GEMMs take for the approximate pass on this shape, beside the fp32 -> fp16 conversion of the query matrix.  Not a
product path: torch.matmul (hipBLASLt / rocBLAS) timings only, to size the idea."""
import json
import torch

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
nq, n, d = 37376, 608, 768
Q = torch.randn((nq, d), device=dev, generator=g)
Q /= Q.norm(dim=1, keepdim=True)
X = torch.randn((n, d), device=dev, generator=g)
X /= X.norm(dim=1, keepdim=True)


def ms(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


out = {}
Xh = X.half()
Qh = Q.half()
S32 = torch.empty((nq, n), device=dev)
Sh = torch.empty((nq, n), device=dev, dtype=torch.half)
out["convert_Q_fp32_to_fp16_us"] = ms(lambda: Qh.copy_(Q))
out["gemm_fp16_in_fp16_out_us"] = ms(lambda: torch.matmul(Qh, Xh.t(), out=Sh))
out["gemm_fp32_sgemm_us"] = ms(lambda: torch.matmul(Q, X.t(), out=S32))
Xb, Qb = X.bfloat16(), Q.bfloat16()
Sb = torch.empty((nq, n), device=dev, dtype=torch.bfloat16)
out["gemm_bf16_us"] = ms(lambda: torch.matmul(Qb, Xb.t(), out=Sb))
# the split form: three fp16 products (hi.hi, hi.lo, lo.hi), fp16 outputs summed in fp32
Ql = ((Q - Qh.float()) * 2048).half()
Xl = ((X - Xh.float()) * 2048).half()
def split3():
    a = torch.matmul(Qh, Xh.t())
    b = torch.matmul(Qh, Xl.t())
    c = torch.matmul(Ql, Xh.t())
    return a, b, c
out["gemm_fp16_x3_us"] = ms(split3)
err_h = (torch.matmul(Qh, Xh.t()).float() - torch.matmul(Q.double(), X.double().t())).abs().max().item()
out["max_abs_err_hi_only_fp16_out"] = err_h
print(json.dumps(out))
