"""Diagnostic (not a test): where the bf16 pipeline's deviation from the fp32 oracle comes from."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_models_gpu import SMALL_UNET, f32_state, rel_l2
from oracle import diffusers_ref as dr
from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
from lbm_amd.models.vae import AutoencoderKL, AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig

BF = torch.bfloat16
cuda = torch.device("cuda:0")
torch.manual_seed(0)
m = DiffusersUNet2DCondWrapper(**SMALL_UNET).to(BF); usd = f32_state(m); m = m.to(cuda).eval()
torch.manual_seed(1)
core = AutoencoderKL(scaling_factor=0.13025).to(BF); vsd = f32_state(core)
w = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version="unused", tiling_size=(128, 128)), vae_model=core).to(cuda).eval()
g = torch.Generator().manual_seed(9)
x = (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).to(BF)
eps = torch.randn(1, 4, 16, 16, generator=g).to(BF)
ts = torch.tensor([1000.0])
with torch.no_grad():
    mom_ref = dr.vae_encode_moments(vsd, x.float())
    z_ref = dr.posterior_sample(mom_ref, eps.float()) * 0.13025
    z = w.encode(x.to(cuda), eps=eps.to(cuda))
    print("encode z rel", rel_l2(z, z_ref), "|z|", float(z_ref.norm()))
    mom = w.vae_model.encode(x.to(cuda)).latent_dist.moments.float().cpu().permute(0, 3, 1, 2)
    print("moments mean rel", rel_l2(mom[:, :4], mom_ref[:, :4]), "logvar rel", rel_l2(mom[:, 4:], mom_ref[:, 4:]))
    pred_ref = dr.unet_forward(usd, SMALL_UNET, z_ref, ts)
    pred_same_in = m(sample=z_ref.to(BF).to(cuda), timestep=ts.to(cuda), conditioning={"cond": {}})
    pred_ref_rounded_in = dr.unet_forward(usd, SMALL_UNET, z_ref.to(BF).float(), ts)
    print("unet rel (same rounded input)", rel_l2(pred_same_in, pred_ref_rounded_in), "|pred|", float(pred_ref.norm()))
    print("oracle sensitivity to input rounding", rel_l2(pred_ref_rounded_in, pred_ref))
    pred = m(sample=z, timestep=ts.to(cuda), conditioning={"cond": {}})
    print("unet rel (pipeline input)", rel_l2(pred, pred_ref))
    lat_ref = z_ref - pred_ref
    lat = (z.float() - pred.float())
    print("latent rel", rel_l2(lat, lat_ref), "|lat|", float(lat_ref.norm()))
    # per-stage VAE encoder
    img_ref = dr.vae_decode(vsd, lat_ref / 0.13025)
    img = w.decode(lat.to(BF))
    print("decode rel", rel_l2(img, img_ref))
    # how far is a plain torch bf16 execution (the reference's own mode) from the fp32 oracle?
    with dr.compute_dtype(BF):
        pred_t16 = dr.unet_forward(usd, SMALL_UNET, z_ref.to(BF), ts)
        print("torch-bf16(cpu) unet rel vs fp32 oracle (rounded input)", rel_l2(pred_t16, pred_ref_rounded_in))
        mom16 = dr.vae_encode_moments(vsd, x)
        print("torch-bf16(cpu) vae moments mean rel", rel_l2(mom16[:, :4], mom_ref[:, :4]))
        img16 = dr.vae_decode(vsd, (lat_ref / 0.13025).to(BF))
        print("torch-bf16(cpu) decode rel", rel_l2(img16, img_ref))
    xx = torch.randn(2, 4, 32, 32, generator=g).to(BF)
    t2 = torch.full((2,), 1000.0)
    r32 = dr.unet_forward(usd, SMALL_UNET, xx.float(), t2)
    with dr.compute_dtype(BF):
        r16 = dr.unet_forward(usd, SMALL_UNET, xx, t2)
    with torch.no_grad():
        rh = m(sample=xx.to(cuda), timestep=t2.to(cuda), conditioning={"cond": {}})
    print("N(0,1) input: hip rel", rel_l2(rh, r32), " torch-bf16 rel", rel_l2(r16, r32))
