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


def test_training_entry_runs_the_shipped_fill_lighting_simple_config(cuda, tmp_path):
    """lbm_amd.trainer.main with the key set of examples/training/config/fill_lighting_simple.yaml and the real builder
    (SD1.5-style UNet, lighting tokens, latent l2 + pixel l2, denoiser and conditioner trained): two optimisation steps
    on an injected loader, a checkpoint per step, resumable by get_model-style loading (model.-prefixed keys)."""
    from lbm_amd.trainer import main
    g = torch.Generator().manual_seed(3)
    R = 128            # latent 16x16: the mid-block attention sees 2x2 = 4 tokens (zero-padded query rows in training)
    batch = {"source": (torch.rand(2, 3, R, R, generator=g) * 2 - 1).to(torch.bfloat16),
             "target": (torch.rand(2, 3, R, R, generator=g) * 2 - 1).to(torch.bfloat16),
             "depth": torch.rand(2, 1, R, R, generator=g).to(torch.bfloat16),
             "lighting_params": torch.rand(2, 8, generator=g)}
    config = dict(
        wandb_project="lbm-fill-lighting-simple", backbone_signature="stable-diffusion-v1-5/stable-diffusion-v1-5",
        unet_input_channels=5, vae_num_channels=4, timestep_sampling="custom_timesteps", selected_timesteps=[250, 500, 750, 1000],
        prob=[0.25, 0.25, 0.25, 0.25], pixel_loss_type="l2", pixel_loss_weight=1.0, latent_loss_type="l2", latent_loss_weight=1.0,
        bridge_noise_sigma=0.005, conditioning_images_keys=[], conditioning_masks_keys=["depth"],
        lighting_embedder_config=dict(mapping_size=256, scale=10.0, hidden_dim=512, out_dim=768, seq_len=5,
                                      param_groups=[[0, 1], [2], [3, 4, 5], [6], [7]]),
        lighting_condition_weight=10.0, concat_condition_weight=1.0, source_key="source", target_key="target",
        train_data_root="unused", validation_data_root="unused", image_size=R, num_workers=0, batch_size=2,
        learning_rate=4e-5, optimizer="AdamW", num_steps=[1, 4], log_interval=500, resume_from_checkpoint=False, max_epochs=2,
        save_interval=1, resume_ckpt_path=None, save_ckpt_path=str(tmp_path))
    torch.manual_seed(0)
    losses = []
    pipe = main("fill_lighting_simple", train_loader=[batch], allow_random_init=True, device=cuda,
                on_step=lambda s, out: losses.append(float(out["loss"])), **config)
    assert pipe.global_step == 2 and len(losses) == 2 and all(np.isfinite(losses))
    groups = pipe.optims[0].groups
    assert len(groups) == 2 and abs(groups[1].lr / groups[0].lr - 10.0) < 1e-9         # conditioner.* at 10x (trainer.py:95)
    assert sum(p.numel() for g_ in groups for p in g_.params) > 859_000_000
    names = sorted(p.name for p in (tmp_path / "fill_lighting_simple").glob("*.ckpt"))
    assert names == ["epoch=0-step=1.ckpt", "epoch=1-step=2.ckpt"]
    ck = torch.load(str(tmp_path / "fill_lighting_simple" / names[-1]), map_location="cpu", weights_only=True)
    assert "model.denoiser.conv_in.weight" in ck["state_dict"] and ck["state_dict"]["model.denoiser.conv_in.weight"].shape[1] == 5
    assert ck["hyper_parameters"]["embedder_1"]["out_dim"] == 768 or ck["hyper_parameters"]["embedder_0"]["out_dim"] == 768
