"""Pixel-space refine model (UNet2DModel [64,128,256,256], 6 -> 3 channels) at full size: ms per 1-NFE sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd.inference.relight import build_filllight_refine_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_filllight_refine_model(backbone_signature="offline", allow_random_init=True, pixel_loss_weight=0.0, bridge_noise_sigma=0.005).to(dev)
src = (torch.rand(B, 3, res, res) * 2 - 1).to(torch.bfloat16).to(dev)
for _ in range(2):
    out = model.sample(src, num_steps=1, conditioner_inputs={"rgb": src})
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = model.sample(src, num_steps=1, conditioner_inputs={"rgb": src})
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert torch.isfinite(out.float()).all()
print(f"refine UNet2DModel {res}^2 batch {B}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} images/s, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
