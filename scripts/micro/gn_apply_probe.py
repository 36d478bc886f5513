"""GroupNorm apply pass (statistics from the producing convolution) with and without SiLU: is it bound by HBM or by the
two transcendentals per element?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best
for (B, H, cin, cout) in [(16, 1024, 128, 128), (16, 512, 256, 256), (16, 256, 512, 512)]:
    x = torch.randn(B, H, H, cin, device=dev, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g).bfloat16() * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    y = ops.conv2d(x, pw, gn_groups=32)
    del x
    gm = torch.ones(cout, device=dev); bt = torch.zeros(cout, device=dev)
    gb = y.numel() * 4 / 1e9
    for silu in (True, False):
        ms = timeit(lambda: ops.groupnorm(y, gm, bt, 32, 1e-6, silu))
        print(f"B{B} {H}^2 C{cout} silu={silu}: {ms:7.3f} ms  {gb/ms:6.2f} TB/s (read + write)", flush=True)
    del y
