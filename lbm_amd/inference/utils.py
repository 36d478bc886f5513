"""Model assembly with the reference's entry points (src/lbm/inference/utils.py:21-222):
``get_model(model_dir, save_dir, torch_dtype, device)`` and ``_get_model_from_config``."""
import logging
import os
from typing import List, Optional

import torch
import yaml

from ..models.embedders import ConditionerWrapper, LatentsConcatEmbedder, LatentsConcatEmbedderConfig
from ..models.lbm import LBMConfig, LBMModel
from ..models.schedulers import FlowMatchEulerDiscreteScheduler
from ..models.unets import DiffusersUNet2DCondWrapper
from ..models.vae import AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig


def get_model(model_dir: str, save_dir: Optional[str] = None, torch_dtype: torch.dtype = torch.bfloat16,
              device: str = "cuda") -> LBMModel:
    if not os.path.exists(model_dir):
        from huggingface_hub import snapshot_download
        model_dir = snapshot_download(model_dir, local_dir=save_dir)

    model_files = os.listdir(model_dir)
    yaml_file = [f for f in model_files if f.endswith(".yaml")]
    if len(yaml_file) == 0:
        raise ValueError("No yaml file found in the model directory.")
    safetensors_files = sorted([f for f in model_files if f.endswith(".safetensors")])
    ckpt_files = sorted([f for f in model_files if f.endswith(".ckpt")])
    if len(safetensors_files) == 0 and len(ckpt_files) == 0:
        raise ValueError("No safetensors or ckpt file found in the model directory")

    with open(os.path.join(model_dir, yaml_file[0]), "r") as f:
        config = yaml.safe_load(f)

    # every weight, the VAE's included, comes from the checkpoint below (strict load): the modules may be built untrained
    config.setdefault("allow_random_init", True)
    model = _get_model_from_config(**config, torch_dtype=torch_dtype)

    if len(safetensors_files) > 0:
        from safetensors.torch import load_file
        logging.info(f"Loading safetensors file: {safetensors_files[-1]}")
        sd = load_file(os.path.join(model_dir, safetensors_files[-1]))
        model.load_state_dict(sd, strict=True)
    else:
        logging.info(f"Loading ckpt file: {ckpt_files[-1]}")
        # weights_only: a Lightning checkpoint's state_dict is plain tensors; nothing else is unpickled
        sd = torch.load(os.path.join(model_dir, ckpt_files[-1]), map_location="cpu", weights_only=True)["state_dict"]
        sd = {k[6:]: v for k, v in sd.items() if k.startswith("model.")}
        model.load_state_dict(sd, strict=True)
    model.to(device).to(torch_dtype)
    model.eval()
    return model


def sdxl_style_unet_kwargs(unet_input_channels: int = 4, vae_num_channels: int = 4) -> dict:
    """The denoiser configuration of src/lbm/inference/utils.py:112-164."""
    return dict(
        in_channels=unet_input_channels, out_channels=vae_num_channels, center_input_sample=False,
        flip_sin_to_cos=True, freq_shift=0,
        down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
        mid_block_type="UNetMidBlock2DCrossAttn",
        up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
        block_out_channels=[320, 640, 1280], layers_per_block=2, downsample_padding=1, act_fn="silu",
        norm_num_groups=32, norm_eps=1e-05, cross_attention_dim=[320, 640, 1280],
        transformer_layers_per_block=[1, 2, 10], attention_head_dim=[5, 10, 20], num_attention_heads=None,
        use_linear_projection=True)


def _get_model_from_config(
    backbone_signature: str = "stabilityai/stable-diffusion-xl-base-1.0",
    vae_num_channels: int = 4,
    unet_input_channels: int = 4,
    timestep_sampling: str = "log_normal",
    selected_timesteps: Optional[List[float]] = None,
    prob: Optional[List[float]] = None,
    conditioning_images_keys: Optional[List[str]] = [],
    conditioning_masks_keys: Optional[List[str]] = [],
    source_key: str = "source_image",
    target_key: str = "source_image_paste",
    bridge_noise_sigma: float = 0.0,
    logit_mean: float = 0.0,
    logit_std: float = 1.0,
    pixel_loss_type: str = "lpips",
    latent_loss_type: str = "l2",
    latent_loss_weight: float = 1.0,
    pixel_loss_weight: float = 0.0,
    torch_dtype: torch.dtype = torch.bfloat16,
    allow_random_init: bool = False,
    **kwargs,
):
    """``backbone_signature`` names where the VAE and the scheduler configuration come from: a local diffusers-layout
    directory, or — with ``allow_random_init=True`` — a known hub name whose VAE architecture is then built with
    untrained weights (no network here; without the flag a hub name raises ``FileNotFoundError``)."""
    conditioners = []
    denoiser = DiffusersUNet2DCondWrapper(**sdxl_style_unet_kwargs(unet_input_channels, vae_num_channels)).to(torch_dtype)

    if conditioning_images_keys != [] or conditioning_masks_keys != []:
        cfg = LatentsConcatEmbedderConfig(image_keys=conditioning_images_keys, mask_keys=conditioning_masks_keys)
        emb = LatentsConcatEmbedder(cfg)
        emb.freeze()
        conditioners.append(emb)
    conditioner = ConditionerWrapper(conditioners=conditioners)

    vae_config = AutoencoderKLDiffusersConfig(version=backbone_signature, subfolder="vae", tiling_size=(128, 128),
                                              allow_random_init=allow_random_init)
    vae = AutoencoderKLDiffusers(vae_config).to(torch_dtype)
    vae.freeze()
    vae.to(torch_dtype)

    config = LBMConfig(
        source_key=source_key, target_key=target_key, latent_loss_weight=latent_loss_weight,
        latent_loss_type=latent_loss_type, pixel_loss_type=pixel_loss_type, pixel_loss_weight=pixel_loss_weight,
        timestep_sampling=timestep_sampling, logit_mean=logit_mean, logit_std=logit_std,
        selected_timesteps=selected_timesteps, prob=prob, bridge_noise_sigma=bridge_noise_sigma)

    sampling_noise_scheduler = FlowMatchEulerDiscreteScheduler.from_pretrained(backbone_signature, subfolder="scheduler",
                                                                               allow_random_init=allow_random_init)

    model = LBMModel(config, denoiser=denoiser, sampling_noise_scheduler=sampling_noise_scheduler, vae=vae,
                     conditioner=conditioner).to(torch_dtype)
    return model
