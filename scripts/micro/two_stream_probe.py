"""Does running the two halves of a batch on two HIP streams overlap the bandwidth-bound passes (GroupNorm apply) of
one half with the MFMA-bound convolutions of the other?  Whole step, batch 16 on one stream against 2 x 8 on two."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
model = bench.build_model(dev, 1024)
B = 16
images = (torch.rand(B, 3, 1024, 1024) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, 128, 128).to(torch.bfloat16).to(dev)
def single():
    return bench.one_step(model, images, eps, 1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = bench.one_step(model, images[:B // 2], eps[:B // 2], 1)
    with torch.cuda.stream(s2):
        b = bench.one_step(model, images[B // 2:], eps[B // 2:], 1)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
def timeit(fn, n=4):
    with torch.no_grad():
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print(f"one stream, batch 16: {timeit(single):7.1f} ms   two streams, 2 x 8: {timeit(dual):7.1f} ms", flush=True)
with torch.no_grad():
    y = single(); a, b = dual()
torch.cuda.synchronize()
print("identical results:", torch.equal(y[:B // 2], a) and torch.equal(y[B // 2:], b))
with torch.no_grad():
    y2 = single(); a2, b2 = dual()
    torch.cuda.synchronize()
    print("single repeatable:", torch.equal(y, y2), " dual repeatable:", torch.equal(a, a2) and torch.equal(b, b2))
    print("first half equal:", torch.equal(y[:B // 2], a), " second half equal:", torch.equal(y[B // 2:], b))
    h = bench.one_step(model, images[:B // 2], eps[:B // 2], 1)
    torch.cuda.synchronize()
    print("batch 8 on the default stream equals batch 16's first half:", torch.equal(h, y[:B // 2]), " equals the two-stream half:", torch.equal(h, a))
    d = (y[:B // 2].float() - a.float()).abs()
    print("max abs diff", float(d.max()), "fraction differing", float((d > 0).float().mean()))
    def psnr(p, q):
        mse = float(((p.float() - q.float()) ** 2).mean())
        return 10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-20))).item()
    print("psnr batch 16 first half vs batch 8: %.1f dB" % psnr(y[:B // 2], h))
