"""Token GEMMs of the 1024^2 batch-16 step (M = 16384 / 65536 tokens, K = 640 ... 5120) over the wide tiles,
back to back (hot caches), with and without a residual operand."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "-1,7,8,10,11,17,18,19").split(",")]
shapes = [(16384, 1280, 1280), (16384, 2560, 1280), (16384, 1280, 5120), (16384, 3840, 1280), (65536, 640, 640), (65536, 1280, 640)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16() * K ** -0.5
    r = torch.randn(M, N, device=dev).bfloat16()
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev).bfloat16())
    fl = 2.0 * M * N * K
    for res in (None, r):
        line = []
        for t in tiles:
            try:
                for _ in range(5):
                    ops.linear(x, pw, tile=t, residual=res)
            except Exception as e:
                line.append(f"t{t}: n/a")
                continue
            best = 1e9
            for rep in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30):
                    ops.linear(x, pw, tile=t, residual=res)
                b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 30)
            line.append(f"t{t}: {best*1e3:6.1f}us {fl/best/1e9:5.0f}TF")
        print(f"M{M} N{N} K{K} res={'y' if res is not None else 'n'}  " + " | ".join(line), flush=True)
