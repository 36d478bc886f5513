"""Cost of the residual operand in the igemm epilogue: each shape with and without it, automatic tile choice and the
named tiles, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best
for (B, H, cin, cout, tiles) in [(4, 1024, 128, 128, (-1, 15, 8)), (4, 1024, 256, 128, (-1,)), (16, 512, 256, 256, (-1, 15, 17)), (16, 256, 512, 512, (-1, 7))]:
    x = torch.randn(B, H, H, cin, device=dev, generator=g).bfloat16()
    r = torch.randn(B, H, H, cout, device=dev, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g).bfloat16() * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    fl = 2.0 * B * H * H * cout * cin * 9
    for t in tiles:
        a = timeit(lambda: ops.conv2d(x, pw, tile=t, gn_groups=32))
        b = timeit(lambda: ops.conv2d(x, pw, tile=t, gn_groups=32, residual=r))
        print(f"conv {cin}->{cout} {H}^2 B{B} t{t} (+stats): {a*1e3:7.1f} us {fl/a/1e9:5.0f} TF | residual {b*1e3:7.1f} us {fl/b/1e9:5.0f} TF  (+{(b-a)*1e3:.1f} us; {B*H*H*cout*2/(b-a)/1e9:.2f} TB/s of residual)", flush=True)
for M, N, K in [(16384, 1280, 1280), (16384, 1280, 5120), (65536, 640, 640), (65536, 640, 2560), (8192, 1280, 1280), (8192, 1280, 5120)]:
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = torch.randn(N, K, device=dev, generator=g).bfloat16() * K ** -0.5
    r = torch.randn(M, N, device=dev, generator=g).bfloat16()
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev).bfloat16())
    fl = 2.0 * M * N * K
    line = []
    for t in (-1, 7, 11, 17, 18):
        a = timeit(lambda: ops.linear(x, pw, tile=t), 30)
        b = timeit(lambda: ops.linear(x, pw, tile=t, residual=r), 30)
        line.append(f"t{t}: {a*1e3:6.1f} / {b*1e3:6.1f} us")
    print(f"M{M} N{N} K{K} (plain / residual)  " + " | ".join(line), flush=True)
