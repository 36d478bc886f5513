"""Generates tests/golden/glue_golden.safetensors and tests/golden/timestep_sampling.json by running the
REFERENCE's own Python (/root/reference/src/lbm) on seeded inputs: the conditioners
(ConditionerWrapper, RawConcatEmbedder, LatentsConcatEmbedder's image path, LightingParamsEmbedder),
LBMModel._get_conditioning with weights != 1, LBMModel._timestep_sampling for its three modes,
LBMModel.forward / sample with conditioners attached (order of the random draws), the pixel loss with
vae=None, and trainer/utils.py's StateDictAdapter / StateDictRenamer.

Runs only in the build container (the reference tree is absent on the GPU box).  Only data is written:
inputs, seeds' resulting draws, module weights (state dicts) and the reference's outputs.

Not covered, because the code path needs a package that is not installed here: LatentsConcatEmbedder's mask
resize (torchvision.transforms.functional.resize).
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import SchedulerShim, _placeholder_modules  # noqa: E402  (also puts the reference on sys.path)


class CondDenoiser(nn.Module):
    """Closed-form stand-in for the UNet that USES its conditioning: concat channels enter through their
    per-pixel channel mean, cross-attention tokens through their per-sample mean."""

    def forward(self, sample, timestep, conditioning=None, **kw):
        t = timestep.float().view(-1, 1, 1, 1) / 1000.0
        out = 0.7 * sample + 0.2 * torch.tanh(sample) * t - 0.1
        cond = (conditioning or {}).get("cond", {})
        if "concat" in cond:
            out = out + 0.05 * cond["concat"].float().mean(dim=1, keepdim=True)
        if "crossattn" in cond:
            out = out + 0.3 * cond["crossattn"].float().mean(dim=(1, 2)).view(-1, 1, 1, 1)
        return out


class ToyVAE(nn.Module):
    """Deterministic 8x-downsampling stand-in with the attributes the conditioners and LBMModel read."""
    downsampling_factor = 8
    latent_channels = 4

    def __init__(self, mix):
        super().__init__()
        self.register_buffer("mix", mix)

    def encode(self, x, batch_size=8):
        return torch.einsum("oc,bchw->bohw", self.mix, torch.nn.functional.avg_pool2d(x, 8))


def main():
    _placeholder_modules()
    from lbm.models.embedders import (ConditionerWrapper, LatentsConcatEmbedder, LatentsConcatEmbedderConfig,
                                      LightingParamsEmbedder, LightingParamsEmbedderConfig, RawConcatEmbedder)
    from lbm.models.lbm import LBMConfig, LBMModel

    out = {}
    meta = {}
    g = torch.Generator().manual_seed(20260211)
    B, L = 3, 16

    # ---- timestep sampling -----------------------------------------------------------------------------------
    ts = {}
    for mode, extra in (("uniform", {}), ("log_normal", dict(logit_mean=0.3, logit_std=1.2)),
                        ("custom_timesteps", dict(selected_timesteps=[250.0, 500.0, 750.0, 1000.0],
                                                  prob=[0.1, 0.2, 0.3, 0.4]))):
        cfg = LBMConfig(timestep_sampling=mode, **extra)
        model = LBMModel(cfg, denoiser=CondDenoiser(), training_noise_scheduler=SchedulerShim(),
                         sampling_noise_scheduler=SchedulerShim())
        for seed in (0, 1, 12345):
            torch.manual_seed(seed)
            np.random.seed(seed)
            t = model._timestep_sampling(n_samples=16)
            ts[f"{mode}.{seed}"] = {"config": extra, "dtype": str(t.dtype), "values": [float(v) for v in t]}
    with open(os.path.join(HERE, "timestep_sampling.json"), "w") as f:
        json.dump(ts, f, indent=1)

    # ---- conditioners ----------------------------------------------------------------------------------------
    batch = {
        "img_a": torch.randn(B, 3, L, L, generator=g),
        "img_b": torch.randn(B, 2, L, L, generator=g),
        "msk": (torch.rand(B, 1, L, L, generator=g) > 0.5).float(),
        "lighting_params": torch.rand(B, 7, generator=g),
        "big": torch.randn(B, 3, 64, 64, generator=g),
    }
    for k, v in batch.items():
        out[f"cond.batch.{k}"] = v
    groups = [[0, 1], [2], [3, 4, 5], [6]]
    torch.manual_seed(5)
    light = LightingParamsEmbedder(LightingParamsEmbedderConfig(mapping_size=16, scale=2.0, hidden_dim=24, out_dim=12,
                                                                seq_len=4, param_groups=groups,
                                                                unconditional_conditioning_rate=0.5))
    for k, v in light.state_dict().items():
        out[f"cond.light_sd.{k}"] = v.clone()
    raw1 = RawConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["img_a"], mask_keys=["msk"], input_key="img_a"))
    raw2 = RawConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["img_b"], mask_keys=None, input_key="img_b",
                                                         unconditional_conditioning_rate=0.25))
    wrapper = ConditionerWrapper([raw1, light, raw2])
    # (a) plain forward: ucg draws for `light` and `raw2` + the lighting embedder's own draw
    torch.manual_seed(77)
    res = wrapper(batch)
    out["cond.wrapper.concat"] = res["cond"]["concat"]
    out["cond.wrapper.crossattn"] = res["cond"]["crossattn"]
    out["cond.wrapper.next_rand"] = torch.rand(1)          # position of the CPU generator after the call
    # (b) ucg rate forced to zero: only the lighting embedder's own draw is left
    torch.manual_seed(77)
    wrapper(batch, set_ucg_rate_zero=True)
    out["cond.wrapper_zero.next_rand"] = torch.rand(1)
    # (c) ucg_keys naming a conditioner: no draw for it
    torch.manual_seed(77)
    wrapper(batch, ucg_keys=["img_b"])
    out["cond.wrapper_keys.next_rand"] = torch.rand(1)

    # LatentsConcatEmbedder, image path only (the mask path needs torchvision)
    mix = torch.randn(4, 3, generator=g)
    out["cond.vae_mix"] = mix
    vae = ToyVAE(mix)
    lat = LatentsConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["big"], mask_keys=None, input_key="big"))
    out["cond.latents_concat.concat"] = lat(batch, vae=vae)["concat"]

    # ---- _get_conditioning with weights, forward and sample with conditioners attached -----------------------
    cfg = LBMConfig(source_key="src", target_key="tgt", timestep_sampling="custom_timesteps",
                    selected_timesteps=[250.0, 500.0, 750.0, 1000.0], prob=[0.25, 0.25, 0.25, 0.25],
                    bridge_noise_sigma=0.005, lighting_condition_weight=0.7, concat_condition_weight=1.3)
    model = LBMModel(cfg, denoiser=CondDenoiser(), training_noise_scheduler=SchedulerShim(),
                     sampling_noise_scheduler=SchedulerShim(), vae=None, conditioner=wrapper)
    torch.manual_seed(31)
    c = model._get_conditioning(batch, set_ucg_rate_zero=True)
    out["getcond.concat"], out["getcond.crossattn"] = c["cond"]["concat"], c["cond"]["crossattn"]
    z = torch.randn(B, 4, L, L, generator=g)
    zs = torch.randn(B, 4, L, L, generator=g)
    out["lbm.z"], out["lbm.z_src"] = z, zs
    fb = dict(batch, src=zs.clone(), tgt=z.clone())
    np.random.seed(3)
    torch.manual_seed(41)
    res = model(fb)
    # replay of the draws, in the reference's order: ucg(light), light debug draw, ucg(raw2) [CPU generator],
    # timestep [numpy], randn_like [CPU generator]
    np.random.seed(3)
    idx = np.random.choice(4, B, p=[0.25] * 4)
    torch.manual_seed(41)
    torch.rand(1), torch.rand(1), torch.rand(1)
    out["lbm.fwd.eps"] = torch.randn_like(z)
    out["lbm.fwd.timestep"] = torch.tensor([250, 500, 750, 1000])[idx].long()
    out["lbm.fwd.loss"] = res["loss"].reshape(1)
    out["lbm.fwd.noisy_sample"] = res["noisy_sample"]
    out["lbm.fwd.predicted_hr"] = res["predicted_hr"]
    torch.manual_seed(43)
    res = model.sample(zs.clone(), num_steps=3, conditioner_inputs=batch, max_samples=2)
    torch.manual_seed(43)
    torch.rand(1)                                           # the lighting embedder's draw (ucg rate forced to zero)
    for i in range(2):
        out[f"lbm.sample3.eps{i}"] = torch.randn(2, 4, L, L)
    out["lbm.sample3.out"] = res

    # ---- pixel loss without a VAE (the prediction is NOT clamped), crop draws included ------------------------
    mask = (torch.rand(B, 1, 24, 24, generator=g) > 0.2).float()
    zp = torch.randn(B, 3, 24, 24, generator=g) * 1.5      # values beyond [-1, 1]
    zsp = torch.randn(B, 3, 24, 24, generator=g) * 1.5
    out["pix.z"], out["pix.z_src"], out["pix.mask"] = zp, zsp, mask
    for kind in ("l2", "l1"):
        cfg = LBMConfig(source_key="src", target_key="tgt", mask_key="mask", timestep_sampling="custom_timesteps",
                        selected_timesteps=[250.0, 500.0, 750.0, 1000.0], prob=[0.25, 0.25, 0.25, 0.25],
                        bridge_noise_sigma=0.005, pixel_loss_type=kind, pixel_loss_weight=2.0, pixel_loss_max_size=16,
                        latent_loss_weight=0.5)
        model = LBMModel(cfg, denoiser=CondDenoiser(), training_noise_scheduler=SchedulerShim(),
                         sampling_noise_scheduler=SchedulerShim(), vae=None, conditioner=None)
        np.random.seed(9)
        torch.manual_seed(51)
        res = model({"src": zsp.clone(), "tgt": zp.clone(), "mask": mask.clone()})
        np.random.seed(9)
        idx = np.random.choice(4, B, p=[0.25] * 4)
        torch.manual_seed(51)
        out[f"pix.{kind}.eps"] = torch.randn_like(zp)
        out[f"pix.{kind}.offsets"] = torch.tensor([torch.randint(0, 8, (1,)).item(), torch.randint(0, 8, (1,)).item()])
        out[f"pix.{kind}.timestep"] = torch.tensor([250, 500, 750, 1000])[idx].long()
        out[f"pix.{kind}.loss"] = res["loss"].reshape(1)
        out[f"pix.{kind}.latent_recon_loss"] = res["latent_recon_loss"].reshape(1)
        out[f"pix.{kind}.pixel_recon_loss"] = res["pixel_recon_loss"].reshape(1)

    # ---- StateDictAdapter / StateDictRenamer (loaded by file path: the package's __init__ needs Lightning) ----
    spec = importlib.util.spec_from_file_location("ref_trainer_utils", "/root/reference/src/lbm/trainer/utils.py")
    ref_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_utils)
    model_sd = {"conv_in.weight": torch.zeros(6, 7, 3, 3), "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight":
                torch.zeros(8, 5), "narrow.weight": torch.zeros(3, 2), "vec.weight": torch.zeros(4, 1),
                "sq.weight": torch.zeros(4), "untouched.weight": torch.zeros(9, 9)}
    ckpt = {"conv_in.weight": torch.randn(6, 4, 3, 3, generator=g),
            "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight": torch.randn(6, 3, generator=g),
            "narrow.weight": torch.randn(5, 4, generator=g), "vec.weight": torch.randn(4, generator=g),
            "sq.weight": torch.randn(4, 3, generator=g), "untouched.weight": torch.randn(2, 2, generator=g)}
    regex = [r"conv_in.weight", r"(down_blocks|up_blocks)\.\d+\.attentions\.\d+\.transformer_blocks\.\d+\.attn\d+\.(to_k|to_v)\.weight",
             r"narrow", r"vec", r"sq"]
    meta["adapter_regex"] = regex
    for k, v in model_sd.items():
        out[f"adapter.model.{k}"] = v
    for k, v in ckpt.items():
        out[f"adapter.ckpt.{k}"] = v
    for strategy in ("zeros", "normal"):
        torch.manual_seed(61)
        res = ref_utils.StateDictAdapter()(model_state_dict=model_sd, checkpoint_state_dict={k: v.clone() for k, v in ckpt.items()},
                                           regex_keys=regex, strategy=strategy)
        for k, v in res.items():
            out[f"adapter.{strategy}.{k}"] = v
    ren = ref_utils.StateDictRenamer()({k: v.clone() for k, v in ckpt.items()},
                                       {"vec.weight": "renamed.weight", "absent": "x"})
    meta["renamer_keys"] = sorted(ren.keys())

    with open(os.path.join(HERE, "glue_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "glue_golden.safetensors"))
    print("wrote", len(out), "tensors")


if __name__ == "__main__":
    main()
