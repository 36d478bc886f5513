"""Does a per-image working set (<= the 256 MB Infinity Cache) make the bandwidth-bound GroupNorm pass and the
convolution that follows faster than the whole batch at once?  (diagnostic)"""
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
for (H, C) in [(1024, 128), (512, 256), (256, 512)]:
    B = 16
    x = torch.randn(B, H, H, C, device=dev, generator=g).to(BF)
    gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g).to(BF) * (9 * C) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(C, device=dev))
    def whole():
        y = ops.conv2d(x, pw, gn_groups=32)
        n = ops.groupnorm(y, gam, bet, 32, 1e-6, True)
        return ops.conv2d(n, pw, gn_groups=32)
    def per_image(k):
        def f():
            outs = []
            for i in range(0, B, k):
                y = ops.conv2d(x[i:i + k], pw, gn_groups=32)
                n = ops.groupnorm(y, gam, bet, 32, 1e-6, True)
                outs.append(ops.conv2d(n, pw, gn_groups=32))
            return outs
        return f
    t_gn = timeit(lambda: ops.groupnorm(x, gam, bet, 32, 1e-6, True))
    t1_gn = timeit(lambda: [ops.groupnorm(x[i:i + 1], gam, bet, 32, 1e-6, True) for i in range(B)])
    print(f"{H}^2 x {C}: groupnorm alone whole {t_gn:.2f} ms, per image {t1_gn:.2f} ms")
    print(f"   conv-gn-conv whole batch {timeit(whole):.2f} ms; per 1 image {timeit(per_image(1)):.2f} ms; per 2 {timeit(per_image(2)):.2f} ms; per 4 {timeit(per_image(4)):.2f} ms")
