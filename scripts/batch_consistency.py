"""Batch consistency / run-to-run determinism of the headline pipeline (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cuda = torch.device("cuda:0")
BF = torch.bfloat16
model = bench.build_model(cuda, 1024)
g = torch.Generator().manual_seed(21)
imgs = (torch.rand(3, 3, 1024, 1024, generator=g) * 2 - 1).to(BF).to(cuda)
eps = torch.randn(3, 4, 128, 128, generator=g).to(BF).to(cuda)
def psnr(a, b):
    mse = float(((a.float() - b.float()) ** 2).mean())
    return 10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-20))).item()
o3a = bench.one_step(model, imgs, eps, 1)
o3b = bench.one_step(model, imgs, eps, 1)
o1 = bench.one_step(model, imgs[1:2].contiguous(), eps[1:2].contiguous(), 1)
print("env", {k: v for k, v in os.environ.items() if k.startswith("LBMK_")})
print("run-to-run identical:", torch.equal(o3a, o3b), " psnr batch3[1] vs alone: %.2f dB" % psnr(o3a[1:2], o1))
