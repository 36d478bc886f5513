"""Split-K for small-M GEMMs as a batched launch over K chunks with f32 partial outputs (diagnostic): GEMM time only,
against the single launch; the reduction pass would add S*M*N*4 bytes of reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, K, N) in [(2048, 10240, 1280), (2048, 3840, 1280), (2048, 1280, 1280), (2048, 5120, 1280), (2048, 1280, 3840), (2048, 1280, 10240),
                  (8192, 1280, 1280), (8192, 5120, 1280), (8192, 640, 640), (8192, 2560, 640), (256, 1280, 1280), (256, 5120, 1280), (1024, 1280, 1280)]:
    x = torch.randn(M, K, device=dev, generator=g).to(BF)
    w = torch.randn(N, K, device=dev, generator=g).to(BF) * K ** -0.5
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev))
    t1 = timeit(lambda: ops.linear(x, pw))
    fl = 2.0 * M * N * K
    row = [f"{M}x{K}->{N}: single {t1:6.1f} us {fl / t1 / 1e6:5.0f} TF |"]
    for S in (2, 4, 8):
        if K % (S * 64): continue
        Kc = K // S
        xv = x.view(M, S, Kc).permute(1, 0, 2)
        wv = w.view(N, S, Kc).permute(1, 0, 2)
        t = timeit(lambda: ops.gemm_nt(xv, wv, out_dtype=torch.float32))
        red = S * M * N * 4 / 4.5e12 * 1e6 + 3
        row.append(f"S={S}: {t:6.1f} us (+~{red:4.1f} reduce) {fl / (t + red) / 1e6:5.0f} TF |")
    print(" ".join(row))
