"""Builders of the relighting model family, with the signatures of the reference
(src/lbm/inference/relight.py:20-417): SD1.5-style UNet (8 heads, 1x1-conv projections,
concat conditioning on VAE latents of shading / normal images, optional lighting tokens).

The reference warm-starts the denoiser from ``StableDiffusionPipeline.from_pretrained`` (hub).
Here ``backbone_signature`` may be a LOCAL directory in the diffusers layout
(``unet/``, ``vae/``, ``scheduler/``): its UNet weights are then adapted exactly like the
reference does (zero-padding ``conv_in`` / ``attn*.to_k|to_v`` to the new shapes, strict load).
A hub name cannot be fetched offline: the builders raise ``FileNotFoundError`` unless ``allow_random_init=True`` is
passed, which builds the architectures with untrained weights (synthetic benchmarks, tests, or training from scratch).
"""
import json
import os
from typing import List, Optional

import torch

from ..models.embedders import (ConditionerWrapper, LatentsConcatEmbedder, LatentsConcatEmbedderConfig,
                                LightingParamsEmbedder, LightingParamsEmbedderConfig, RawConcatEmbedder)
from ..models.lbm import LBMConfig, LBMModel
from ..models.schedulers import FlowMatchEulerDiscreteScheduler
from ..models.unets import DiffusersUNet2DCondWrapper
from ..models.vae import AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig
from ..trainer.utils import StateDictAdapter

_ADAPT_KEYS = [
    r"conv_in.weight",
    r"(down_blocks|up_blocks)\.\d+\.attentions\.\d+\.transformer_blocks\.\d+\.attn\d+\.(to_k|to_v)\.weight",
    r"mid_block\.attentions\.\d+\.transformer_blocks\.\d+\.attn\d+\.(to_k|to_v)\.weight",
]


def sd15_style_unet_kwargs(unet_input_channels: int, vae_num_channels: int, cross_attention_dim) -> dict:
    """The denoiser configuration of src/lbm/inference/relight.py:61-120."""
    return dict(
        in_channels=unet_input_channels, out_channels=vae_num_channels, center_input_sample=False,
        flip_sin_to_cos=True, freq_shift=0,
        down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
        mid_block_type="UNetMidBlock2DCrossAttn",
        up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
        block_out_channels=[320, 640, 1280, 1280], layers_per_block=2, downsample_padding=1, act_fn="silu",
        norm_num_groups=32, norm_eps=1e-05, cross_attention_dim=cross_attention_dim,
        transformer_layers_per_block=1, attention_head_dim=8, num_attention_heads=None, use_linear_projection=False)


def _local_unet_state(backbone_signature: str):
    p = os.path.join(backbone_signature, "unet")
    if not os.path.isdir(p):
        return None, None
    from safetensors.torch import load_file
    files = sorted(f for f in os.listdir(p) if f.endswith(".safetensors"))
    if not files:
        return None, None
    cfg = {}
    if os.path.exists(os.path.join(p, "config.json")):
        with open(os.path.join(p, "config.json")) as f:
            cfg = json.load(f)
    return load_file(os.path.join(p, files[0])), cfg


def _build_relight_model(
        backbone_signature: str = "stable-diffusion-v1-5/stable-diffusion-v1-5",
        vae_num_channels: int = 4,
        unet_input_channels: int = 12,
        timestep_sampling: str = "log_normal",
        selected_timesteps: Optional[List[float]] = None,
        prob: Optional[List[float]] = None,
        conditioning_images_keys: Optional[List[str]] = None,
        conditioning_masks_keys: Optional[List[str]] = None,
        lighting_conditioning: bool = False,
        lighting_embedder_config: Optional[dict] = None,
        source_key: str = "source",
        target_key: str = "target",
        mask_key: Optional[str] = None,
        bridge_noise_sigma: float = 0.0,
        logit_mean: float = 0.0,
        logit_std: float = 1.0,
        pixel_loss_type: str = "lpips",
        latent_loss_type: str = "l2",
        latent_loss_weight: float = 1.0,
        pixel_loss_weight: float = 0.0,
        cross_attention_dim=None,
        lighting_condition_weight: float = 1.0,
        concat_condition_weight: float = 1.0,
        allow_random_init: bool = False,
) -> LBMModel:
    if conditioning_images_keys is None:
        conditioning_images_keys = ["shading", "normal"]
    if conditioning_masks_keys is None:
        conditioning_masks_keys = []

    ckpt_sd, ckpt_cfg = _local_unet_state(backbone_signature)
    if cross_attention_dim is None:
        cross_attention_dim = (ckpt_cfg or {}).get("cross_attention_dim", 768)   # SD1.5 text width

    denoiser = DiffusersUNet2DCondWrapper(
        **sd15_style_unet_kwargs(unet_input_channels, vae_num_channels, cross_attention_dim)).to(torch.bfloat16)
    if ckpt_sd is not None:
        for key in ("add_embedding.linear_1.weight", "add_embedding.linear_1.bias",
                    "add_embedding.linear_2.weight", "add_embedding.linear_2.bias"):
            ckpt_sd.pop(key, None)
        ckpt_sd = StateDictAdapter()(model_state_dict=denoiser.state_dict(), checkpoint_state_dict=ckpt_sd,
                                     regex_keys=_ADAPT_KEYS, strategy="zeros")
        denoiser.load_state_dict(ckpt_sd, strict=True)
    elif not allow_random_init:
        raise FileNotFoundError(
            f"'{backbone_signature}' is not a local diffusers-layout directory (unet/ vae/ scheduler/) and hub downloads are "
            "unavailable: the denoiser cannot be warm-started. Pass a local directory, or allow_random_init=True to keep "
            "the UNTRAINED initialisation (synthetic runs, training from scratch).")

    conditioners = []
    if conditioning_images_keys != [] or conditioning_masks_keys != []:
        emb = LatentsConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=conditioning_images_keys,
                                                                mask_keys=conditioning_masks_keys))
        emb.freeze()
        conditioners.append(emb)
    if lighting_conditioning:
        conditioners.append(LightingParamsEmbedder(LightingParamsEmbedderConfig(**(lighting_embedder_config or {}))))
    conditioner = ConditionerWrapper(conditioners=conditioners)

    vae = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version=backbone_signature, subfolder="vae",
                                                              tiling_size=(128, 128), allow_random_init=allow_random_init))
    vae.freeze()
    vae.to(torch.bfloat16)

    config = LBMConfig(
        source_key=source_key, target_key=target_key, mask_key=mask_key, latent_loss_weight=latent_loss_weight,
        latent_loss_type=latent_loss_type, pixel_loss_type=pixel_loss_type, pixel_loss_weight=pixel_loss_weight,
        timestep_sampling=timestep_sampling, logit_mean=logit_mean, logit_std=logit_std,
        selected_timesteps=selected_timesteps, prob=prob, bridge_noise_sigma=bridge_noise_sigma,
        lighting_condition_weight=lighting_condition_weight, concat_condition_weight=concat_condition_weight)

    model = LBMModel(
        config, denoiser=denoiser,
        training_noise_scheduler=FlowMatchEulerDiscreteScheduler.from_pretrained(
            backbone_signature, subfolder="scheduler", allow_random_init=allow_random_init),
        sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler.from_pretrained(
            backbone_signature, subfolder="scheduler", allow_random_init=allow_random_init),
        vae=vae, conditioner=conditioner).to(torch.bfloat16)
    return model


def build_relight_model(backbone_signature: str = "stable-diffusion-v1-5/stable-diffusion-v1-5", **kwargs) -> LBMModel:
    """attn2 of every block is a self-attention: cross_attention_dim equals the block's channels
    (relight.py:269) and no crossattn conditioning is produced."""
    kwargs.setdefault("lighting_conditioning", False)
    return _build_relight_model(backbone_signature=backbone_signature, cross_attention_dim=[320, 640, 1280, 1280], **kwargs)


def build_filllight_model(backbone_signature: str = "stable-diffusion-v1-5/stable-diffusion-v1-5", **kwargs) -> LBMModel:
    """Lighting tokens (LightingParamsEmbedder) drive a real cross-attention of width 768."""
    kwargs.setdefault("lighting_conditioning", True)
    return _build_relight_model(backbone_signature=backbone_signature, **kwargs)


def build_filllight_refine_model(
        backbone_signature: str = "stable-diffusion-v1-5/stable-diffusion-v1-5",
        unet_input_channels: int = 6,
        unet_output_channels: int = 3,
        block_out_channels: Optional[List[int]] = None,
        layers_per_block: int = 2,
        timestep_sampling: str = "log_normal",
        selected_timesteps: Optional[List[float]] = None,
        prob: Optional[List[float]] = None,
        conditioning_images_keys: Optional[List[str]] = None,
        conditioning_masks_keys: Optional[List[str]] = None,
        source_key: str = "source",
        target_key: str = "target",
        mask_key: Optional[str] = None,
        bridge_noise_sigma: float = 0.0,
        logit_mean: float = 0.0,
        logit_std: float = 1.0,
        pixel_loss_type: str = "lpips",
        pixel_loss_weight: float = 1.0,
        latent_loss_type: str = "l2",
        latent_loss_weight: float = 1.0,
        allow_random_init: bool = False,
) -> LBMModel:
    """Pixel-space refine model (relight.py:325-417): UNet2DModel [64, 128, 256, 256] on 6-channel inputs
    (bridge sample + raw rgb conditioning), no VAE."""
    from ..models.unets import DiffusersUNet2DWrapper
    if conditioning_images_keys is None:
        conditioning_images_keys = ["rgb"]
    if conditioning_masks_keys is None:
        conditioning_masks_keys = []
    if block_out_channels is None:
        block_out_channels = [64, 128, 256, 256]
    denoiser = DiffusersUNet2DWrapper(
        in_channels=unet_input_channels, out_channels=unet_output_channels,
        down_block_types=["DownBlock2D"] * len(block_out_channels), up_block_types=["UpBlock2D"] * len(block_out_channels),
        block_out_channels=block_out_channels, layers_per_block=layers_per_block, norm_num_groups=32, act_fn="silu",
        center_input_sample=False).to(torch.bfloat16)
    conditioners = []
    if conditioning_images_keys or conditioning_masks_keys:
        emb = RawConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=conditioning_images_keys, mask_keys=conditioning_masks_keys))
        emb.freeze()
        conditioners.append(emb)
    config = LBMConfig(
        ucg_keys=None, source_key=source_key, target_key=target_key, mask_key=mask_key,
        latent_loss_weight=latent_loss_weight, latent_loss_type=latent_loss_type, pixel_loss_type=pixel_loss_type,
        pixel_loss_weight=pixel_loss_weight, timestep_sampling=timestep_sampling, logit_mean=logit_mean, logit_std=logit_std,
        selected_timesteps=selected_timesteps, prob=prob, bridge_noise_sigma=bridge_noise_sigma)
    return LBMModel(
        config, denoiser=denoiser,
        training_noise_scheduler=FlowMatchEulerDiscreteScheduler.from_pretrained(
            backbone_signature, subfolder="scheduler", allow_random_init=allow_random_init),
        sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler.from_pretrained(
            backbone_signature, subfolder="scheduler", allow_random_init=allow_random_init),
        vae=None, conditioner=ConditionerWrapper(conditioners=conditioners)).to(torch.bfloat16)
