"""Small-GEMM tile comparison: y[M,N] = x[M,K] w[N,K]^T for the token counts of a 512^2 training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "-1,0,1,3,4,6,9,10").split(",")]
shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 10240, 1280), (8192, 640, 640), (8192, 640, 2560), (2048, 3840, 1280)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16() * K ** -0.5
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev).bfloat16())
    fl = 2.0 * M * N * K
    for t in tiles:
        for _ in range(10):
            ops.linear(x, pw, tile=t)
    line = []
    for t in tiles:
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                ops.linear(x, pw, tile=t)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 50)
        line.append(f"t{t}: {best*1e3:6.1f}us {fl/best/1e9:5.0f}TF")
    print(f"M{M} N{N} K{K}  " + " | ".join(line), flush=True)
