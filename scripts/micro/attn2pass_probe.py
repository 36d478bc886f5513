"""Pieces of the two-pass wide-head attention against the unfused path (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
B, N, C = 2, 16384, 512
q = torch.randn(B, N, C, device=dev, generator=g).to(BF); k = torch.randn(B, N, C, device=dev, generator=g).to(BF)
vt = torch.randn(B, C, N, device=dev, generator=g).to(BF)
scale = C ** -0.5; alpha = scale * 1.4426950408889634
parts = ops.smx_parts(N)
part = torch.empty(B, N, parts, dtype=torch.float32, device=dev)
p = torch.empty(B, N, N, dtype=BF, device=dev)
ops.gemm_nt(q, k, alpha=alpha, smx=(1, None, part)); rowmax = part.amax(-1)
print("pass 1 (row maxima)      %.3f ms" % timeit(lambda: ops.gemm_nt(q, k, alpha=alpha, smx=(1, None, part))))
print("amax                     %.3f ms" % timeit(lambda: part.amax(-1)))
print("pass 2 (exp, bf16 store) %.3f ms" % timeit(lambda: ops.gemm_nt(q, k, alpha=alpha, out=p, smx=(2, rowmax, part))))
print("P V (f32 out)            %.3f ms" % timeit(lambda: ops.gemm_nt(p, vt, out_dtype=torch.float32)))
print("P V (bf16 out)           %.3f ms" % timeit(lambda: ops.gemm_nt(p, vt)))
o = ops.gemm_nt(p, vt, out_dtype=torch.float32); inv = part.sum(-1).reciprocal_()
print("sum + scale + cast       %.3f ms" % timeit(lambda: (o * part.sum(-1).reciprocal_().unsqueeze(-1)).to(BF)))
print("two-pass total           %.3f ms" % timeit(lambda: ops.attention_two_pass(q, k, vt, scale)))
def old():
    s = ops.gemm_nt(q, k, out_dtype=torch.float32, alpha=scale)
    return ops.gemm_nt(ops.softmax_rows(s), vt)
print("unfused total            %.3f ms" % timeit(old))
print("  scores f32             %.3f ms" % timeit(lambda: ops.gemm_nt(q, k, out_dtype=torch.float32, alpha=scale)))
s = ops.gemm_nt(q, k, out_dtype=torch.float32, alpha=scale)
print("  softmax rows           %.3f ms" % timeit(lambda: ops.softmax_rows(s)))
