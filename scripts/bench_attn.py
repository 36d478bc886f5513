"""Micro-benchmark of the flash-attention forward kernels on the benchmark's attention shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for (B, H, N, D) in [(16, 10, 4096, 64), (16, 20, 1024, 64), (2, 8, 16384, 40), (4, 8, 4096, 80), (8, 8, 1024, 160)]:
    q = torch.randn(B, N, H * D, device=dev, generator=g).bfloat16()
    k = torch.randn(B, N, H * D, device=dev, generator=g).bfloat16()
    vt = torch.randn(B, H * D, N, device=dev, generator=g).bfloat16()
    ms = timeit(lambda: ops.flash_attn(q, k, vt, H, D))
    fl = 4.0 * B * H * N * N * D
    print(f"B{B} H{H} N{N} D{D}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
