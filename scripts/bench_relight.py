"""SD1.5-style relight model (src/lbm/inference/relight.py: 865 M-parameter UNet, 12 input channels = latent + VAE
latents of the shading and normal images): encode source + two conditioning images, 1-NFE sample, decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd.inference.relight import build_relight_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nfe = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_relight_model(backbone_signature="offline", allow_random_init=True, unet_input_channels=12, bridge_noise_sigma=0.005).to(torch.bfloat16).to(dev)
model.eval()
g = torch.Generator().manual_seed(1)
imgs = {k: (torch.rand(B, 3, res, res, generator=g) * 2 - 1).to(torch.bfloat16).to(dev) for k in ("source", "shading", "normal")}

@torch.no_grad()
def step():
    z = model.vae.encode(imgs["source"])
    return model.sample(z, num_steps=nfe, conditioner_inputs=imgs)

for _ in range(2):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
assert out.shape == (B, 3, res, res) and torch.isfinite(out.float()).all()
print(f"relight (SD1.5-style UNet) {res}^2 batch {B} {nfe}-NFE: {dt * 1e3:.1f} ms/step, {B / dt:.2f} images/s, "
      f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
