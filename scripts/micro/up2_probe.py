"""Upsampling convolution: gather-through-upsample 3x3 kernel vs four 2x2 parity convolutions (diagnostic)."""
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
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 7, 17, 8]
for (B, H, C) in [(16, 128, 512), (16, 256, 512), (16, 512, 256)]:
    x = torch.randn(B, H, H, C, device=dev, generator=g).to(BF)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g).to(BF) * (9 * C) ** -0.5
    bias = torch.zeros(C, device=dev)
    pw = ops.pack_conv_weight(w, bias)
    parts = ops.pack_conv_weight_up2(w, bias)
    fl9 = 2.0 * B * 4 * H * H * C * 9 * C
    t9 = timeit(lambda: ops.conv2d(x, pw, up2=True, gn_groups=32))
    print(f"up2 conv {C}->{C} in {H}^2 B{B}: 3x3 through upsample {t9:.2f} ms ({fl9 / t9 / 1e9:.0f} TF)")
    for t in tiles:
        t4 = timeit(lambda: ops.conv2d_up2(x, parts, gn_groups=32, tile=t))
        print(f"    four 2x2 tile {t:3d}: {t4:.2f} ms ({fl9 * 4 / 9 / t4 / 1e9:.0f} TF executed)")
    xl = torch.randn(B, H, H, C, device=dev, generator=g).to(BF)
    pw1 = parts[3][0]
    # one parity as a plain (unscattered) 2x2 convolution, for the cost of the scatter
    d_plain = timeit(lambda: ops.conv2d(xl, pw1, pad=(0, 0), out_hw=(H, H)))
    print(f"    one plain 2x2 conv (no scatter): {d_plain:.2f} ms ({2.0 * B * H * H * C * 4 * C / d_plain / 1e9:.0f} TF)")
