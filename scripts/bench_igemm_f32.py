"""fp32-mode implicit GEMM (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) on the shapes of the 512x512 batch-1 step.
python scripts/bench_igemm_f32.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("convolutions (3x3, stride 1): H  Cin Cout   us   TFLOP/s")
for H, cin, cout in ((512, 128, 128), (512, 256, 128), (256, 256, 256), (128, 512, 512), (64, 512, 512), (64, 320, 320), (32, 640, 640),
                     (16, 1280, 1280), (16, 2560, 1280)):
    x = torch.randn(1, H, H, cin, device=dev, generator=g)
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.02
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    t = timed(lambda: ops.conv2d(x, pw, stride=1, pad=(1, 1)))
    fl = 2.0 * H * H * cout * cin * 9
    print(f"   {H:5d} {cin:5d} {cout:5d} {t*1e6:9.1f} {fl/t/1e12:7.1f}", flush=True)
print("linear layers: M  N  K   us   TFLOP/s")
for M, N, K in ((4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120),
                (1024, 3840, 1280), (16384, 1280, 1280)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    pw = ops.pack_linear_weight(w, None)
    t = timed(lambda: ops.linear(x, pw))
    fl = 2.0 * M * N * K
    print(f"   {M:6d} {N:6d} {K:6d} {t*1e6:9.1f} {fl/t/1e12:7.1f}", flush=True)
