"""Generates tests/golden/*.safetensors by running the REFERENCE's own Python
(/root/reference/src/lbm: LBMModel.forward / LBMModel.sample / Tiler) on seeded inputs.

Runs only in the build container (the reference tree is not present on the GPU box);
only tensors are written — inputs, the explicit random draws, and the reference's outputs.

The reference imports lpips / diffusers / torchvision at module import time; none is
installed here.  None of their arithmetic is used by the paths exercised below (the
denoiser is a closed-form toy function, vae=None), so empty placeholder modules providing
the imported NAMES are registered first.  The scheduler handed to LBMModel is the
restatement in oracle/diffusers_ref.py (diffusers itself being unavailable), so these
vectors pin the reference-owned arithmetic: interpolant, loss, sampling loop, tiling.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")


def _placeholder_modules():
    class _Any(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("lpips", LPIPS=_Any)
    d = mod("diffusers", FlowMatchEulerDiscreteScheduler=_Any)
    d.schedulers = mod("diffusers.schedulers", FlowMatchEulerDiscreteScheduler=_Any)
    d.models = mod("diffusers.models", UNet2DConditionModel=_Any, UNet2DModel=_Any, AutoencoderKL=_Any)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")


class SchedulerShim:
    """Adapter giving oracle.diffusers_ref.FlowMatchEulerOracle the call signature LBMModel uses."""

    def __init__(self):
        from oracle.diffusers_ref import FlowMatchEulerOracle
        self._o = FlowMatchEulerOracle()
        self.config = types.SimpleNamespace(num_train_timesteps=1000)

    @property
    def timesteps(self):
        return self._o.timesteps

    @property
    def sigmas(self):
        return self._o.sigmas

    def set_timesteps(self, sigmas=None, **kw):
        self._o.set_timesteps(sigmas)

    def step(self, model_output, timestep, sample, return_dict=False):
        return (self._o.step(model_output, timestep, sample),)


class ToyDenoiser(nn.Module):
    """Deterministic closed-form stand-in for the UNet: pred = a*x + b*tanh(x)*t/1000 + c."""

    def forward(self, sample, timestep, conditioning=None, **kw):
        t = timestep.float().view(-1, 1, 1, 1) / 1000.0
        return 0.7 * sample + 0.2 * torch.tanh(sample) * t - 0.1


def main():
    _placeholder_modules()
    from lbm.models.lbm import LBMConfig, LBMModel
    from lbm.models.utils import Tiler, pad

    out = {}
    g = torch.Generator().manual_seed(20260130)
    B, C, L = 4, 4, 16
    z = torch.randn(B, C, L, L, generator=g)
    zs = torch.randn(B, C, L, L, generator=g)
    mask = (torch.rand(B, 1, L, L, generator=g) > 0.25).float()

    # ---- training forward: custom timesteps, l2 / l1 / masked --------------------------
    for name, loss_type, use_mask in (("l2", "l2", False), ("l1", "l1", False), ("masked", "l2", True)):
        cfg = LBMConfig(source_key="src", target_key="tgt", mask_key="mask" if use_mask else None,
                        latent_loss_type=loss_type, timestep_sampling="custom_timesteps",
                        selected_timesteps=[250.0, 500.0, 750.0, 1000.0], prob=[0.25, 0.25, 0.25, 0.25],
                        bridge_noise_sigma=0.005)
        model = LBMModel(cfg, denoiser=ToyDenoiser(), training_noise_scheduler=SchedulerShim(),
                         sampling_noise_scheduler=SchedulerShim(), vae=None, conditioner=None)
        np.random.seed(7)
        torch.manual_seed(11)
        batch = {"src": zs.clone(), "tgt": z.clone()}
        if use_mask:
            batch["mask"] = mask.clone()
        res = model(batch)
        # replay the draws the reference made, in its order: timestep (numpy), then randn_like (torch)
        np.random.seed(7)
        idx = np.random.choice(4, B, p=[0.25] * 4)
        torch.manual_seed(11)
        eps = torch.randn_like(z)
        out[f"fwd_{name}.timestep"] = torch.tensor([250, 500, 750, 1000])[idx].long()
        out[f"fwd_{name}.eps"] = eps
        out[f"fwd_{name}.loss"] = res["loss"].reshape(1)
        out[f"fwd_{name}.noisy_sample"] = res["noisy_sample"]
        out[f"fwd_{name}.predicted_hr"] = res["predicted_hr"]
    out["fwd.z"], out["fwd.z_src"], out["fwd.mask"] = z, zs, mask

    # ---- sampling: 1 and 4 steps ---------------------------------------------------------
    cfg = LBMConfig(source_key="src", target_key="tgt", bridge_noise_sigma=0.005)
    model = LBMModel(cfg, denoiser=ToyDenoiser(), training_noise_scheduler=SchedulerShim(),
                     sampling_noise_scheduler=SchedulerShim(), vae=None, conditioner=None)
    for n in (1, 4):
        torch.manual_seed(23)
        res = model.sample(zs.clone(), num_steps=n)
        torch.manual_seed(23)
        for i in range(n - 1):
            out[f"sample_{n}.eps{i}"] = torch.randn_like(zs)
        out[f"sample_{n}.out"] = res

    # ---- scheduler tables -------------------------------------------------------------------
    for n in (1, 2, 4, 20):
        s = SchedulerShim()
        s.set_timesteps(sigmas=np.linspace(1, 1 / n, n))
        out[f"sched_{n}.sigmas"] = s.sigmas.clone()
        out[f"sched_{n}.timesteps"] = s.timesteps.clone()

    # ---- Tiler: 40x36 latent, tile 16, overlap 4, scale 2 (ragged last tiles) ----------------
    zt = torch.randn(1, 4, 40, 36, generator=g)
    proj = torch.randn(3, 4, generator=g)

    def toy_decode(t):   # x2 nearest upsample + channel mix: position dependent, cheap
        up = torch.nn.functional.interpolate(t, scale_factor=2.0, mode="nearest")
        return torch.einsum("oc,bchw->bohw", proj, up)

    tiler = Tiler()
    tiles = tiler.get_tiles(input=zt, tile_size=(16, 16), overlap_size=(4, 4), scale=2, out_channels=3)
    for i, row in enumerate(tiles):
        for j, tile in enumerate(row):
            shp = tile.shape
            dec = toy_decode(pad(tile, base_h=16, base_w=16))
            tiles[i][j] = dec[0, :, : shp[2] * 2, : shp[3] * 2].unsqueeze(0)
    merged = tiler.merge_tiles(tiles=tiles)
    out["tiler.z"], out["tiler.proj"], out["tiler.merged"] = zt, proj, merged.float()
    out["tiler.weights_24x32"] = tiler._gaussian_weights(24, 32, 1, 1)[0, 0].float()

    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "bridge_golden.safetensors"))
    print("wrote", len(out), "tensors")


if __name__ == "__main__":
    main()
