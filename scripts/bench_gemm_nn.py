"""Input gradient of a linear layer, dX = dY . W: the untransposed-weight kernel (ops.gemm_nn) against the implicit-GEMM
kernel on a W^T pack (ops.linear on pack_linear_weight(W^T)), per row count.  python scripts/bench_gemm_nn.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print(f"{'M':>6} {'N':>6} {'K':>6}   gemm_nn us  TF/s    igemm(W^T) us  TF/s")
for M in (2048, 8192, 32768):
    for N, K in ((1280, 1280), (10240, 1280), (1280, 5120), (3840, 1280), (640, 640), (5120, 640), (640, 2560)):
        dy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(torch.bfloat16)
        pt = ops.pack_linear_weight(ops.transpose(w), None)
        fl = 2.0 * M * N * K
        a = timed(lambda: ops.gemm_nn(dy, w))
        b = timed(lambda: ops.linear(dy, pt))
        print(f"{M:6d} {N:6d} {K:6d}   {a*1e6:9.1f} {fl/a/1e12:6.0f}    {b*1e6:9.1f} {fl/b/1e12:6.0f}", flush=True)
