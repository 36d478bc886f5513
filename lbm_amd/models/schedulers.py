"""Flow-matching Euler scheduler with the interface the reference uses from diffusers'
``FlowMatchEulerDiscreteScheduler`` (built at src/lbm/inference/utils.py:209-212 and
src/lbm/inference/relight.py:202-209; used at src/lbm/models/lbm/lbm_model.py:279-286,
301-302, 320, 346): ``timesteps``, ``sigmas``, ``config.num_train_timesteps``,
``set_timesteps(sigmas=)`` and ``step(pred, t, sample, return_dict=False)[0]``.

The schedule tables stay on the host (a handful of floats); ``step`` launches the Euler
kernel, which does the f32 update and the cast back to the prediction dtype in one pass.
"""
from types import SimpleNamespace

import numpy as np
import torch

from .. import ops


class FlowMatchEulerDiscreteScheduler:
    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, **unused):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sigmas = torch.from_numpy(ts) / num_train_timesteps
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self.sigmas = sigmas
        self._step_index = None
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    @classmethod
    def from_pretrained(cls, name_or_path=None, subfolder=None, allow_random_init: bool = False, **kw):
        """A local ``<dir>/<subfolder>/scheduler_config.json`` is read for ``num_train_timesteps`` / ``shift``.  The
        scheduler JSONs of the Stable Diffusion 1.x / SDXL repositories (the backbones the reference names) describe
        other scheduler classes and contribute ``num_train_timesteps = 1000`` only, so those names give the default
        schedule without any download.  Anything else raises ``FileNotFoundError`` (no network to look it up) unless
        ``allow_random_init`` is set, which accepts the default schedule."""
        import json
        import os
        if name_or_path and os.path.isdir(name_or_path):
            p = os.path.join(name_or_path, subfolder or "", "scheduler_config.json")
            if os.path.exists(p):
                with open(p) as f:
                    cfg = json.load(f)
                return cls(num_train_timesteps=cfg.get("num_train_timesteps", 1000), shift=cfg.get("shift", 1.0))
        name = str(name_or_path or "").lower()
        if name_or_path is None or allow_random_init or "stable-diffusion" in name:
            return cls()
        raise FileNotFoundError(f"scheduler config '{name_or_path}' is neither a local directory holding "
                                f"{subfolder or '.'}/scheduler_config.json nor a known Stable Diffusion backbone name, and "
                                "hub downloads are unavailable; pass allow_random_init=True to take the default schedule")

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, **kw):
        if sigmas is None:
            t = np.linspace(self.config.num_train_timesteps * self.sigma_max,
                            self.config.num_train_timesteps * self.sigma_min, num_inference_steps)
            sigmas = t / self.config.num_train_timesteps
        s = np.array(sigmas).astype(np.float32)
        s = self.config.shift * s / (1 + (self.config.shift - 1) * s)
        s = torch.from_numpy(s)
        self.timesteps = s * self.config.num_train_timesteps
        self.sigmas = torch.cat([s, torch.zeros(1)])
        self._step_index = None

    def index_for_timestep(self, timestep):
        idx = (self.timesteps == (timestep.cpu() if torch.is_tensor(timestep) else timestep)).nonzero()
        return idx[1 if len(idx) > 1 else 0].item()

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True,
             eps: torch.Tensor = None, bridge_noise_sigma: float = 0.0, **kw):
        """prev = cast(sample + (sigma_next - sigma) * model_output).  When ``eps`` is given the
        bridge re-noising of lbm_model.py:347-351 is fused into the same kernel."""
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        s = float(self.sigmas[self._step_index])
        s_next = float(self.sigmas[self._step_index + 1])
        x = sample.to(model_output.dtype).contiguous()
        prev = ops.euler_step(x, model_output.contiguous(), s, s_next, eps=eps, bridge_sigma=bridge_noise_sigma)
        self._step_index += 1
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)
