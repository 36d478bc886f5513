"""LBMModel.forward / sample with conditioners attached, the conditioning weights, and the pixel loss without a VAE,
on the GPU, against tensors produced by the reference's own LBMModel + conditioners
(tests/golden/glue_golden.safetensors, written by tests/golden/make_golden_glue.py).  fp32 bridge kernels: 1e-6."""
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from test_glue_golden import build_wrapper, sub

pytestmark = pytest.mark.gpu
GOLD = load_file(os.path.join(os.path.dirname(__file__), "golden", "glue_golden.safetensors"))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cond_denoiser(sample, timestep, conditioning=None, **kw):
    t = timestep.float().view(-1, 1, 1, 1) / 1000.0
    out = 0.7 * sample + 0.2 * torch.tanh(sample) * t - 0.1
    cond = (conditioning or {}).get("cond", {})
    if "concat" in cond:
        out = out + 0.05 * cond["concat"].float().mean(dim=1, keepdim=True)
    if "crossattn" in cond:
        out = out + 0.3 * cond["crossattn"].float().mean(dim=(1, 2)).view(-1, 1, 1, 1)
    return out


def scheds():
    from lbm_amd.models.schedulers import FlowMatchEulerDiscreteScheduler
    return dict(training_noise_scheduler=FlowMatchEulerDiscreteScheduler(),
                sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler())


def test_forward_and_sample_with_conditioners_match_reference(cuda):
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    cfg = LBMConfig(source_key="src", target_key="tgt", timestep_sampling="custom_timesteps",
                    selected_timesteps=[250.0, 500.0, 750.0, 1000.0], prob=[0.25, 0.25, 0.25, 0.25],
                    bridge_noise_sigma=0.005, lighting_condition_weight=0.7, concat_condition_weight=1.3)
    model = LBMModel(cfg, denoiser=cond_denoiser, vae=None, conditioner=build_wrapper(), **scheds()).to(cuda)
    batch = {k: v.to(cuda) for k, v in sub("cond.batch.").items()}
    torch.manual_seed(31)
    c = model._get_conditioning(batch, set_ucg_rate_zero=True)
    assert c["cond"]["concat"].is_cuda and rel_l2(c["cond"]["concat"], GOLD["getcond.concat"]) < 1e-6
    assert rel_l2(c["cond"]["crossattn"], GOLD["getcond.crossattn"]) < 1e-5
    z, zs = GOLD["lbm.z"].to(cuda), GOLD["lbm.z_src"].to(cuda)
    # the timestep is drawn by the model itself (numpy generator), after the conditioners' CPU draws
    np.random.seed(3)
    torch.manual_seed(41)
    out = model(dict(batch, src=zs, tgt=z), bridge_eps=GOLD["lbm.fwd.eps"].to(cuda))
    assert rel_l2(out["noisy_sample"], GOLD["lbm.fwd.noisy_sample"]) < 1e-6
    assert rel_l2(out["predicted_hr"], GOLD["lbm.fwd.predicted_hr"]) < 1e-5
    assert rel_l2(out["loss"].reshape(1), GOLD["lbm.fwd.loss"]) < 1e-5
    # sampling: 3 steps, conditioning and latents cut to max_samples = 2
    eps_list = [GOLD[f"lbm.sample3.eps{i}"].to(cuda) for i in range(2)]
    res = model.sample(zs, num_steps=3, conditioner_inputs=batch, max_samples=2, eps_list=eps_list)
    assert res.shape == GOLD["lbm.sample3.out"].shape
    assert rel_l2(res, GOLD["lbm.sample3.out"]) < 1e-6


@pytest.mark.parametrize("kind", ["l2", "l1"])
def test_pixel_loss_without_vae_matches_reference(cuda, kind):
    """vae=None: the prediction is compared unclamped (lbm_model.py:235-236); crop offsets come from the CPU generator
    after the bridge noise would have been drawn, so they are injected here together with the noise"""
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    cfg = LBMConfig(source_key="src", target_key="tgt", mask_key="mask", timestep_sampling="custom_timesteps",
                    selected_timesteps=[250.0, 500.0, 750.0, 1000.0], prob=[0.25, 0.25, 0.25, 0.25],
                    bridge_noise_sigma=0.005, pixel_loss_type=kind, pixel_loss_weight=2.0, pixel_loss_max_size=16,
                    latent_loss_weight=0.5)
    model = LBMModel(cfg, denoiser=cond_denoiser, vae=None, conditioner=None, **scheds()).to(cuda)
    z, zs, mask = GOLD["pix.z"].to(cuda), GOLD["pix.z_src"].to(cuda), GOLD["pix.mask"].to(cuda)
    # CPU generator replay: randn_like is a device draw here, so consume the reference's CPU draw first; the crop
    # offsets then come out of the same generator state the reference had
    torch.manual_seed(51)
    torch.randn(z.shape)
    out = model({"src": zs, "tgt": z, "mask": mask}, bridge_eps=GOLD[f"pix.{kind}.eps"].to(cuda),
                timesteps=GOLD[f"pix.{kind}.timestep"])
    assert rel_l2(out["pixel_recon_loss"].reshape(1), GOLD[f"pix.{kind}.pixel_recon_loss"]) < 1e-5
    assert rel_l2(out["latent_recon_loss"].reshape(1), GOLD[f"pix.{kind}.latent_recon_loss"]) < 1e-5
    assert rel_l2(out["loss"].reshape(1), GOLD[f"pix.{kind}.loss"]) < 1e-5
