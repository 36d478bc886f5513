"""LBMModel with the reference's interface (src/lbm/models/lbm/lbm_model.py:17-408):
``forward(batch) -> dict`` (bridge-matching training loss) and
``sample(z, num_steps, conditioner_inputs, max_samples) -> Tensor`` (Euler bridge sampler).

The bridge arithmetic runs in the HIP kernels (interpolant, Euler step + re-noise, latent
loss); the Python control flow keeps the reference's order of random draws.  Keyword-only
additions (default ``None`` = draw exactly like the reference):

  * ``forward(..., bridge_eps=, timesteps=)`` — the noise of lbm_model.py:143 and the
    timesteps of :134;
  * ``sample(..., eps_list=)`` — the per-step noise of lbm_model.py:350.

``loss.backward()`` runs the HIP backward kernels (lbm_amd/autograd.py), the pixel loss (l2 / l1 / LPIPS through
the frozen VAE decoder, lbm_model.py:172-175, 197-249) included.
"""
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ... import ops, streams
from ..base.base_model import BaseModel
from ..schedulers import FlowMatchEulerDiscreteScheduler
from .lbm_config import LBMConfig


class LBMModel(BaseModel):
    @classmethod
    def load_from_config(cls, config: LBMConfig):
        return cls(config=config)

    def __init__(self, config: LBMConfig, denoiser=None, training_noise_scheduler=None,
                 sampling_noise_scheduler=None, vae=None, conditioner=None):
        BaseModel.__init__(self, config)
        self.vae = vae
        self.denoiser = denoiser
        self.conditioner = conditioner
        self.sampling_noise_scheduler = sampling_noise_scheduler
        self.training_noise_scheduler = training_noise_scheduler
        self.timestep_sampling = config.timestep_sampling
        self.latent_loss_type = config.latent_loss_type
        self.latent_loss_weight = config.latent_loss_weight
        self.pixel_loss_type = config.pixel_loss_type
        self.pixel_loss_max_size = config.pixel_loss_max_size
        self.pixel_loss_weight = config.pixel_loss_weight
        self.logit_mean = config.logit_mean
        self.logit_std = config.logit_std
        self.prob = config.prob
        self.selected_timesteps = config.selected_timesteps
        self.source_key = config.source_key
        self.target_key = config.target_key
        self.mask_key = config.mask_key
        self.bridge_noise_sigma = config.bridge_noise_sigma
        self.num_iterations = nn.Parameter(torch.tensor(0, dtype=torch.float32), requires_grad=False)
        if self.pixel_loss_type == "lpips" and self.pixel_loss_weight > 0:
            from ..lpips import LPIPS
            self.lpips_loss = LPIPS(net="vgg")           # lbm_model.py:82-83; weights come with the checkpoint
        else:
            self.lpips_loss = None

    def on_fit_start(self, device: torch.device | None = None, *args, **kwargs):
        super().on_fit_start(device=device, *args, **kwargs)
        if self.vae is not None:
            self.vae.on_fit_start(device=device, *args, **kwargs)
        if self.conditioner is not None:
            self.conditioner.on_fit_start(device=device, *args, **kwargs)

    # ---- training loss ----------------------------------------------------------------------
    def forward(self, batch: Dict[str, Any], step=0, batch_idx=0, *args, bridge_eps: Optional[torch.Tensor] = None,
                timesteps: Optional[torch.Tensor] = None, **kwargs):
        self.num_iterations += 1
        if self.vae is not None:
            z = self.vae.encode(batch[self.target_key])
            downsampling_factor = self.vae.downsampling_factor
        else:
            z = batch[self.target_key]
            downsampling_factor = 1

        if self.mask_key in batch:
            valid_mask = batch[self.mask_key].bool()[:, 0, :, :].unsqueeze(1)
            valid_mask_pixels = valid_mask
            valid_mask_for_latent = ~torch.max_pool2d((~valid_mask).float(), downsampling_factor, downsampling_factor).bool()
            valid_mask_for_latent = valid_mask_for_latent.repeat((1, z.shape[1], 1, 1))
        else:
            valid_mask_for_latent = None          # all valid
            valid_mask_pixels = None

        source_image = batch[self.source_key]
        z_source = self.vae.encode(source_image) if self.vae is not None else source_image

        conditioning = self._get_conditioning(batch, *args, **kwargs)

        timestep = timesteps if timesteps is not None else self._timestep_sampling(n_samples=z.shape[0], device=z.device)
        timestep = timestep.to(z.device)

        sched = self.training_noise_scheduler
        sigmas = self._get_sigmas(sched, timestep, n_dim=4, device=z.device)
        if bridge_eps is None:
            bridge_eps = torch.randn_like(z)
        t_cpu = timestep.detach().cpu()
        copy_src = torch.tensor([bool(t.item() == sched.timesteps[0]) for t in t_cpu], device=z.device)
        z, z_source = z.contiguous(), z_source.contiguous()
        noisy_sample = ops.bridge_interp(z_source, z, bridge_eps.to(z.dtype).contiguous(), sigmas, self.bridge_noise_sigma,
                                         copy_src=copy_src, out_dtype=torch.float32)

        prediction = self.denoiser(sample=noisy_sample, timestep=timestep, conditioning=conditioning, *args, **kwargs)

        neg_sig = -sigmas.reshape(-1)
        denoised_sample = ops.axpy_rows(noisy_sample, prediction.detach().contiguous(), neg_sig)

        if self.latent_loss_weight > 0:
            if self.latent_loss_type not in ("l2", "l1"):
                raise NotImplementedError(f"Loss type {self.latent_loss_type} not implemented")
            if prediction.requires_grad:
                # training: the loss is differentiable w.r.t. the prediction (kernel lbmk_latent_loss_bwd)
                from ...autograd import LatentLossFn
                loss = LatentLossFn.apply(prediction.contiguous(), z_source, z, valid_mask_for_latent,
                                          self.latent_loss_type == "l1")
            else:
                loss = ops.latent_loss(prediction.contiguous(), z_source, z, valid_mask_for_latent,
                                       l1=self.latent_loss_type == "l1")
            latent_recon_loss = loss.mean()
        else:
            loss = torch.zeros(z.shape[0], device=z.device)
            latent_recon_loss = torch.zeros_like(loss)
        if self.pixel_loss_weight > 0:
            if prediction.requires_grad:
                from ...autograd import AxpyRowsFn
                x0 = AxpyRowsFn.apply(noisy_sample, prediction, neg_sig)
            else:
                x0 = denoised_sample
            pixel_loss = self.pixel_loss(x0, batch[self.target_key].detach(), valid_mask_pixels)
            loss = loss + self.pixel_loss_weight * pixel_loss
        else:
            pixel_loss = torch.zeros_like(latent_recon_loss)

        return {"loss": loss.mean(), "latent_recon_loss": latent_recon_loss, "pixel_recon_loss": pixel_loss.mean(),
                "predicted_hr": denoised_sample, "noisy_sample": noisy_sample}

    def latent_loss(self, prediction, model_input, valid_latent_mask):
        """Same signature as the reference (lbm_model.py:186-195): model_input is the target z_src - z."""
        if self.latent_loss_type not in ("l2", "l1"):
            raise NotImplementedError(f"Loss type {self.latent_loss_type} not implemented")
        zeros = torch.zeros_like(model_input)
        return ops.latent_loss(prediction.contiguous(), model_input.contiguous(), zeros, valid_latent_mask,
                               l1=self.latent_loss_type == "l1")

    def pixel_loss(self, prediction, model_input, valid_mask, offsets=None):
        """Random crop to ``pixel_loss_max_size``, decode, clamp, l2 / l1 / LPIPS against the target pixels
        (lbm_model.py:197-249; crop offsets are drawn on the host in the reference's order).
        ``offsets=(offset_h, offset_w)`` overrides the draw (tests)."""
        f = self.vae.downsampling_factor if self.vae is not None else 1
        latent_crop, input_crop = self.pixel_loss_max_size // f, self.pixel_loss_max_size
        crop_h = max(prediction.shape[2] - latent_crop, 0)
        crop_w = max(prediction.shape[3] - latent_crop, 0)
        input_crop_h = max(model_input.shape[2] - input_crop, 0)
        input_crop_w = max(model_input.shape[3] - input_crop, 0)
        offset_h = 0 if crop_h == 0 else int(torch.randint(0, crop_h, (1,)).item())
        offset_w = 0 if crop_w == 0 else int(torch.randint(0, crop_w, (1,)).item())
        if offsets is not None:
            offset_h, offset_w = offsets
        ioh, iow = offset_h * f, offset_w * f
        prediction = prediction[:, :, crop_h - offset_h: min(crop_h - offset_h + latent_crop, prediction.shape[2]),
                                crop_w - offset_w: min(crop_w - offset_w + latent_crop, prediction.shape[3])]
        sl_h = slice(input_crop_h - ioh, min(input_crop_h - ioh + input_crop, model_input.shape[2]))
        sl_w = slice(input_crop_w - iow, min(input_crop_w - iow + input_crop, model_input.shape[3]))
        model_input = model_input[:, :, sl_h, sl_w].contiguous()
        if valid_mask is not None:
            valid_mask = valid_mask[:, :, sl_h, sl_w]
        # only a VAE-decoded prediction is clamped (lbm_model.py:233-236): the pixel-space refine model (no VAE)
        # compares, and back-propagates through, the raw prediction
        clamp = self.vae is not None
        decoded = self.vae.decode(prediction) if clamp else prediction
        if self.pixel_loss_type == "lpips":
            # lpips_loss(decoded * mask, target * mask).mean() (lbm_model.py:248-249); clamp and mask are fused
            # into the network's input stage
            if self.lpips_loss is None:
                raise NotImplementedError("pixel_loss_type 'lpips' needs pixel_loss_weight > 0 at construction")
            return self.lpips_loss(decoded, model_input, mask=valid_mask, clamp0=clamp).mean()
        if self.pixel_loss_type not in ("l2", "l1"):
            raise NotImplementedError(f"pixel loss type {self.pixel_loss_type} not implemented")
        from ...autograd import PixelLossFn
        return PixelLossFn.apply(decoded.contiguous(), model_input, valid_mask, self.pixel_loss_type == "l1", clamp)

    def _get_conditioning(self, batch: Dict[str, Any], ucg_keys: List[str] = None, set_ucg_rate_zero=False, *args, **kwargs):
        """Conditioner outputs with the two per-kind weights of the config applied (lbm_model.py:251-275): cross-attention
        tokens times ``lighting_condition_weight``, concatenated maps times ``concat_condition_weight`` (a weight of 1 leaves
        the tensor untouched, so no multiplication and no new tensor)."""
        if self.conditioner is None:
            return None
        out = self.conditioner(batch, *args, ucg_keys=ucg_keys, set_ucg_rate_zero=set_ucg_rate_zero, vae=self.vae, **kwargs)
        cond = out.get("cond") if out else None
        if cond is not None:
            for kind, weight in (("crossattn", self.config.lighting_condition_weight),
                                 ("concat", self.config.concat_condition_weight)):
                if cond.get(kind) is not None and weight != 1.0:
                    cond[kind] = cond[kind] * weight
        return out

    def _timestep_sampling(self, n_samples=1, device="cpu"):
        """One training timestep per sample, drawn on the HOST from the generators the reference uses, in its order
        (lbm_model.py:277-291): torch's CPU generator for "uniform" (one randint) and "log_normal" (one normal), numpy's
        global generator for "custom_timesteps" (one choice with the configured probabilities)."""
        mode = self.timestep_sampling
        if mode == "custom_timesteps":
            picks = np.random.choice(len(self.selected_timesteps), n_samples, p=self.prob)
            return torch.tensor(self.selected_timesteps, device=device, dtype=torch.long)[picks]
        sched = self.training_noise_scheduler
        n_train = sched.config.num_train_timesteps
        if mode == "uniform":
            picks = torch.randint(0, n_train, (n_samples,), device="cpu")
        elif mode == "log_normal":
            logits = torch.normal(mean=self.logit_mean, std=self.logit_std, size=(n_samples,), device="cpu")
            picks = (torch.sigmoid(logits) * n_train).long()
        else:
            return None          # LBMConfig validates the mode; the reference falls through the same way
        return sched.timesteps[picks].to(device=device)

    def _predicted_x_0(self, model_output, sample, sigmas=None):
        return ops.axpy_rows(sample.contiguous(), model_output.contiguous(), -sigmas.reshape(-1))

    def _get_sigmas(self, scheduler, timesteps, n_dim=4, dtype=torch.float32, device="cpu"):
        """Exact-match lookup of each timestep in the schedule, on the host: the tables are a few
        floats, so no device round trip per sample (the reference does B ``.item()`` syncs, :304)."""
        sched_t = scheduler.timesteps.detach().cpu()
        sig = scheduler.sigmas.detach().cpu().to(dtype)
        t_cpu = timesteps.detach().cpu()
        idx = [(sched_t == t).nonzero().item() for t in t_cpu]
        sigma = sig[idx].flatten().to(device)
        while len(sigma.shape) < n_dim:
            sigma = sigma.unsqueeze(-1)
        return sigma

    # ---- sampling ---------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, z: torch.Tensor, num_steps: int = 20, conditioner_inputs: Optional[Dict[str, Any]] = None,
               max_samples: Optional[int] = None, verbose: bool = False, *, eps_list: Optional[List[torch.Tensor]] = None):
        sched = self.sampling_noise_scheduler
        sched.set_timesteps(sigmas=np.linspace(1, 1 / num_steps, num_steps))
        sample = z
        conditioning = self._get_conditioning(conditioner_inputs, set_ucg_rate_zero=True, device=z.device)
        if max_samples is not None:
            sample = sample[:max_samples]
        if conditioning:
            conditioning["cond"] = {k: v[:max_samples] for k, v in conditioning["cond"].items()}
        sample = sample.contiguous()
        fused = isinstance(sched, FlowMatchEulerDiscreteScheduler)
        n = len(sched.timesteps)
        for i, t in enumerate(sched.timesteps):
            denoiser_input = sched.scale_model_input(sample, t) if hasattr(sched, "scale_model_input") else sample
            # the timestep vector is filled on the device from the host scalar: no host-to-device copy in the
            # loop, so the whole call can be captured into a HIP graph
            t_vec = torch.full((denoiser_input.shape[0],), float(t), dtype=sched.timesteps.dtype, device=z.device)
            pred = self._denoise_batch(denoiser_input, t_vec, conditioning)
            last = i == n - 1
            eps = None
            if not last:
                eps = eps_list[i].to(sample) if eps_list is not None else torch.randn_like(sample)
            if fused:
                # Euler step and bridge re-noise in one kernel (f32 math, cast to the latent dtype)
                sample = sched.step(pred, t, sample, return_dict=False, eps=eps,
                                    bridge_noise_sigma=self.bridge_noise_sigma)[0]
            else:
                sample = sched.step(pred, t, sample, return_dict=False)[0]
                if not last:
                    s_next = float(sched.sigmas[i + 1])
                    sample = ops.euler_step(sample.contiguous(), sample.contiguous(), s_next, s_next, eps=eps,
                                            bridge_sigma=self.bridge_noise_sigma)
            sample = sample.to(z.dtype)
        return self.vae.decode(sample) if self.vae is not None else sample

    def _denoise_batch(self, x: torch.Tensor, t_vec: torch.Tensor, conditioning) -> torch.Tensor:
        """One denoiser evaluation; batches of 16 or more run as two halves on two streams (lbm_amd.streams: measured -1.9 % on the 1024^2 batch-16 step, nothing at 8)."""
        n = x.shape[0]
        if not streams.enabled(x) or n < 16:
            return self.denoiser(sample=x, timestep=t_vec, conditioning=conditioning)
        h = (n + 1) // 2

        def part(lo, hi):
            cond = conditioning
            if conditioning:
                cond = {k: ({kk: vv[lo:hi] for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in conditioning.items()}
            return (x[lo:hi], t_vec[lo:hi], cond)

        outs = streams.run_slices(lambda p: self.denoiser(sample=p[0], timestep=p[1], conditioning=p[2]),
                                  (part(0, h), part(h, n)), device=x.device)
        return torch.cat(outs, dim=0)

    def log_samples(self, batch: Dict[str, Any], input_shape: Optional[Tuple[int, int, int]] = None,
                    max_samples: Optional[int] = None, num_steps: Union[int, List[int]] = 20):
        if isinstance(num_steps, int):
            num_steps = [num_steps]
        logs = {}
        N = max_samples if max_samples is not None else len(batch[self.source_key])
        batch = {k: v[:N] for k, v in batch.items()}
        if input_shape is None and self.vae is None:
            raise ValueError("input_shape must be passed when no VAE is used in the model")
        for num_step in num_steps:
            source_image = torch.nn.functional.interpolate(batch[self.source_key], size=batch[self.target_key].shape[2:],
                                                           mode="bilinear", align_corners=False).to(dtype=self.dtype)
            z = self.vae.encode(source_image) if self.vae is not None else source_image
            logs[f"samples_{num_step}_steps"] = self.sample(z, num_steps=num_step, conditioner_inputs=batch, max_samples=N)
        return logs
