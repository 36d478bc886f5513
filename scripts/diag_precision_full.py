"""Diagnostic (not a test): error budget of the full-size bf16 pipeline (2.43 B-parameter SDXL-style UNet, SD VAE,
1024x1024, 1 step) against the fp32 oracle: encode / UNet on identical inputs / the UNet's sensitivity to the encode
error / the sampled latent.  python scripts/diag_precision_full.py [L]   (latent side, default 128)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from lbm_amd.inference.utils import sdxl_style_unet_kwargs
from oracle import diffusers_ref as dr

BF = torch.bfloat16
L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cuda = torch.device("cuda:0")
torch.set_num_threads(min(16, os.cpu_count() or 1))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


model = bench.build_model(cuda, 8 * L)
usd = {k: v.detach().float().cpu() for k, v in model.denoiser.state_dict().items()}
vsd = {k: v.detach().float().cpu() for k, v in model.vae.vae_model.state_dict().items()}
cfg = sdxl_style_unet_kwargs()
g = torch.Generator().manual_seed(5)
x = (torch.rand(1, 3, 8 * L, 8 * L, generator=g) * 2 - 1).to(BF)
eps = torch.randn(1, 4, L, L, generator=g).to(BF)
ts = torch.tensor([1000.0])
with torch.no_grad():
    t0 = time.time()
    z_ref = dr.posterior_sample(dr.vae_encode_moments(vsd, x.float()), eps.float()) * 0.13025
    z = model.vae.encode(x.to(cuda), eps=eps.to(cuda))
    print(f"encode: rel {rel(z, z_ref):.5f}  |z| {float(z_ref.norm()):.2f}   ({time.time() - t0:.0f} s)", flush=True)
    zr = z_ref.to(BF)
    pred_ref = dr.unet_forward(usd, cfg, z_ref, ts)
    pred_ref_r = dr.unet_forward(usd, cfg, zr.float(), ts)
    pred_ref_z = dr.unet_forward(usd, cfg, z.float().cpu(), ts)
    print(f"|pred| {float(pred_ref.norm()):.2f}  |z - pred| {float((z_ref - pred_ref).norm()):.2f}", flush=True)
    print(f"oracle sensitivity: to bf16 rounding of z {rel(pred_ref_r, pred_ref):.5f}, to the HIP encode's z {rel(pred_ref_z, pred_ref):.5f}")
    p_same = model.denoiser(sample=zr.to(cuda), timestep=ts.to(cuda), conditioning={"cond": {}})
    print(f"unet, identical (rounded) input: hip rel {rel(p_same, pred_ref_r):.5f}")
    p_pipe = model.denoiser(sample=z, timestep=ts.to(cuda), conditioning={"cond": {}})
    print(f"unet, pipeline input: rel to oracle(z_ref) {rel(p_pipe, pred_ref):.5f}; rel to oracle(hip z) {rel(p_pipe, pred_ref_z):.5f}")
    lat_ref = z_ref - pred_ref
    lat = z.float().cpu() - p_pipe.float().cpu()
    print(f"latent: rel {rel(lat, lat_ref):.5f}  (bf16-rounded result {rel(lat.to(BF), lat_ref):.5f})")
    lat_u = z_ref - p_same.float().cpu()
    print(f"latent with exact encode, hip unet: rel {rel(lat_u, lat_ref):.5f}")
    lat_e = z.float().cpu() - pred_ref_z
    print(f"latent with hip encode, exact unet: rel {rel(lat_e, lat_ref):.5f}")
    if os.environ.get("DIAG_T16"):
        with dr.compute_dtype(BF):
            p16 = dr.unet_forward(usd, cfg, zr, ts)
        print(f"torch-bf16 (CPU) unet, identical input: rel {rel(p16, pred_ref_r):.5f}")
    # the same UNet on a unit-variance latent (the module-level tests' input)
    xn = torch.randn(1, 4, L, L, generator=g).to(BF)
    rn = dr.unet_forward(usd, cfg, xn.float(), ts)
    hn = model.denoiser(sample=xn.to(cuda), timestep=ts.to(cuda), conditioning={"cond": {}})
    print(f"unet on N(0,1) input: hip rel {rel(hn, rn):.5f}")
    if os.environ.get("DIAG_T16"):
        with dr.compute_dtype(BF):
            n16 = dr.unet_forward(usd, cfg, xn, ts)
        print(f"torch-bf16 (CPU) unet on N(0,1) input: rel {rel(n16, rn):.5f}")
