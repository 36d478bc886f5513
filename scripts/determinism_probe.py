"""Probe: is one inference step bit-reproducible run to run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B, res = 2, 1024
dev = torch.device("cuda:0")
model = bench.build_model(dev, res)
images = (torch.rand(B, 3, res, res) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, res // 8, res // 8).to(torch.bfloat16).to(dev)
with torch.no_grad():
    z1 = model.vae.encode(images, eps=eps); z2 = model.vae.encode(images, eps=eps)
    print("encode equal:", torch.equal(z1, z2), float((z1.float() - z2.float()).abs().max()))
    t = torch.full((B,), 1000.0, device=dev)
    p1 = model.denoiser(sample=z1, timestep=t, conditioning={"cond": {}}); p2 = model.denoiser(sample=z1, timestep=t, conditioning={"cond": {}})
    print("unet equal:", torch.equal(p1, p2), float((p1.float() - p2.float()).abs().max()))
    d1 = model.vae.decode(z1); d2 = model.vae.decode(z1)
    print("decode equal:", torch.equal(d1, d2), float((d1.float() - d2.float()).abs().max()))
