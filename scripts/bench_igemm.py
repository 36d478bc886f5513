"""Micro-benchmark of the implicit-GEMM tile shapes on the benchmark's dominant layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g).to(BF)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 5, 6]
cases = []
for (B, H, cin, cout) in [(4, 1024, 128, 128), (4, 1024, 256, 128), (8, 512, 256, 256), (16, 256, 512, 512),
                          (16, 128, 512, 512), (16, 128, 320, 320), (16, 128, 960, 320), (16, 64, 640, 640),
                          (16, 32, 1280, 1280), (16, 32, 2560, 1280)]:
    x = rnd(B, H, H, cin)
    w = rnd(cout, cin, 3, 3) * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    cases.append((f"conv3x3 B{B} {H}^2 {cin}->{cout}", 2.0 * B * H * H * cout * 9 * cin, lambda t, x=x, pw=pw: ops.conv2d(x, pw, tile=t)))
for (M, K, N, geglu) in [(16384, 1280, 10240, True), (16384, 5120, 1280, False), (16384, 1280, 2560, False),
                         (16384, 1280, 1280, False), (65536, 640, 5120, True), (65536, 640, 640, False), (65536, 2560, 640, False)]:
    x = rnd(M, K)
    w = rnd(N, K) * K ** -0.5
    pw = ops.pack_geglu_weight(w, torch.zeros(N, device=dev)) if geglu else ops.pack_linear_weight(w, torch.zeros(N, device=dev))
    cases.append((f"linear {M}x{K}->{N}{' geglu' if geglu else ''}", 2.0 * M * N * K, lambda t, x=x, pw=pw: ops.linear(x, pw, tile=t)))

print(f"{'case':40s} " + " ".join(f"tile{t:>2d} TF/s" for t in tiles))
for name, fl, fn in cases:
    row = []
    for t in tiles:
        try:
            ms = timeit(lambda: fn(t))
            row.append(f"{fl / ms / 1e9:10.0f}")
        except Exception as e:
            row.append(f"{'err':>10s}")
    print(f"{name:40s} " + " ".join(row))
