"""Micro-benchmark of GroupNorm(+SiLU), LayerNorm and the row softmax on the benchmark's shapes (GB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for (B, H, C) in [(8, 1024, 128), (8, 512, 256), (8, 512, 128), (8, 256, 512), (8, 128, 512), (16, 128, 320), (16, 64, 640),
                  (16, 32, 1280), (16, 32, 2560), (16, 64, 1920), (16, 128, 960)]:
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True))
    gb = x.numel() * 2 * 2 / 1e9    # algorithmic: one read + one write
    print(f"groupnorm+silu B{B} {H}^2 C{C}: {ms:8.3f} ms  {gb / ms * 1e3:7.0f} GB/s algorithmic (r+w)")
for (rows, C) in [(16384, 1280), (65536, 640)]:
    x = torch.randn(rows, C, device=dev).bfloat16()
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.layernorm(x, g, b))
    print(f"layernorm {rows}x{C}: {ms:8.3f} ms  {x.numel() * 4 / ms / 1e6:7.0f} GB/s")
s = torch.randn(4, 16384, 16384, device=dev)
ms = timeit(lambda: ops.softmax_rows(s), iters=5)
print(f"softmax_rows 4x16384x16384 f32->bf16: {ms:8.3f} ms  {s.numel() * 6 / ms / 1e6:7.0f} GB/s")
