"""Weight-gradient GEMM: the untransposed-operand kernel against transposes + NT GEMM, per linear-layer shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 10240, 1280), (2048, 3840, 1280), (2048, 5120, 1280),
          (8192, 640, 640), (8192, 640, 2560), (8192, 5120, 640), (8192, 1920, 640), (32768, 320, 320)]
def timeit(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N, K in shapes:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    old = timeit(lambda: ops.gemm_tn_t(ops.transpose(dy), ops.transpose(x), out=out))
    res = {}
    for s in (None, 1, 2, 3, 4, 6, 8):
        try:
            res[s] = timeit(lambda: ops.gemm_tn(dy, x, out=out, splits=s))
        except Exception as e:
            res[s] = float("nan")
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: transposes+NT {old:7.1f} us | TN " + " ".join(f"s{k}:{v:6.1f}" for k, v in res.items())
          + f" | best {fl / min(v for v in res.values() if v == v) / 1e6:5.0f} TF", flush=True)
