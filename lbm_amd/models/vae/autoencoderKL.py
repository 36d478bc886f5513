"""VAE wrapper with the reference's interface (src/lbm/models/vae/autoencoderKL.py:9-136):
``encode(x, batch_size=8)``, ``decode(z)``, ``downsampling_factor``, ``latent_channels``,
``vae_model``.  Differences that are deliberate and documented:

  * ``encode`` / ``decode`` accept explicit ``eps`` / return tensors on the GPU: the tiled
    decode keeps every tile and the Gaussian-weighted merge on the device (the reference
    moves each tile to the CPU and merges there, autoencoderKL.py:124-129);
  * the down-sampling factor is read off the architecture instead of probing a random
    32x32 encode at construction time (autoencoderKL.py:54-58), so construction consumes
    no RNG draws.
"""
from typing import List, Optional

import torch

from ... import ops, streams
from ..base.base_model import BaseModel
from ..utils import Tiler, gaussian_weights_1d, tile_starts
from .autoencoder_kl import AutoencoderKL, DiagonalGaussianDistribution
from .autoencoderKL_config import AutoencoderKLDiffusersConfig


class AutoencoderKLDiffusers(BaseModel):
    def __init__(self, config: AutoencoderKLDiffusersConfig, vae_model: Optional[AutoencoderKL] = None):
        BaseModel.__init__(self, config)
        self.config = config
        self.vae_model = vae_model if vae_model is not None else AutoencoderKL.from_pretrained(
            config.version, subfolder=config.subfolder, revision=config.revision, allow_random_init=config.allow_random_init)
        self.tiling_size = config.tiling_size
        self.tiling_overlap = config.tiling_overlap
        self._get_properties()

    def _get_properties(self):
        cfg = self.vae_model.config
        self.has_shift_factor = getattr(cfg, "shift_factor", None) is not None
        self.shift_factor = cfg.shift_factor if self.has_shift_factor else 0
        self.latent_channels = cfg.latent_channels
        self.has_latents_mean = getattr(cfg, "latents_mean", None) is not None
        self.has_latents_std = getattr(cfg, "latents_std", None) is not None
        self.latents_mean = getattr(cfg, "latents_mean", None)
        self.latents_std = getattr(cfg, "latents_std", None)
        self.downsampling_factor = self.vae_model.downsampling_factor

    def encode(self, x: torch.Tensor, batch_size: int = 8, eps: Optional[torch.Tensor] = None,
               generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """(posterior sample - shift) * scaling_factor, in chunks of ``batch_size`` images
        (autoencoderKL.py:60-69).  ``eps``: optional NCHW noise for the posterior sample."""
        latents: List[torch.Tensor] = []
        chunks = [x[i:i + batch_size] for i in range(0, x.shape[0], batch_size)]
        if len(chunks) > 1 and streams.enabled(x):
            # the chunks are independent: alternate them between two streams (the posterior draws stay in chunk order)
            moments = streams.run_slices(lambda c: self.vae_model.encode(c).latent_dist.moments, chunks)
            dists = [DiagonalGaussianDistribution(m, self.vae_model.config.latent_channels, x.dtype) for m in moments]
        else:
            dists = [self.vae_model.encode(c).latent_dist for c in chunks]
        for k, dist in enumerate(dists):
            e = None if eps is None else eps[k * batch_size:(k + 1) * batch_size]
            latents.append(dist.sample(generator=generator, eps=e, shift=float(self.shift_factor),
                                       scale=float(self.vae_model.config.scaling_factor)))
        return latents[0] if len(latents) == 1 else torch.cat(latents, dim=0)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        if self.has_latents_mean and self.has_latents_std:
            raise NotImplementedError("latents_mean / latents_std un-scaling is not implemented (absent from the SD / SDXL VAE configs)")
        if torch.is_grad_enabled() and z.requires_grad:
            # differentiable route (pixel loss): gradients flow to the latent through the frozen decoder
            from ...autograd import NchwToNhwcFn, NhwcToNchwFn, ScaleShiftFn
            if z.shape[2] > self.tiling_size[0] or z.shape[3] > self.tiling_size[1]:
                raise NotImplementedError("tiled decode has no backward; crop the latent to the tile size (pixel_loss_max_size)")
            zs = ScaleShiftFn.apply(z.contiguous().to(torch.bfloat16), 1.0 / float(self.vae_model.config.scaling_factor),
                                    float(self.shift_factor))
            img = self.vae_model.decode_nhwc(NchwToNhwcFn.apply(zs))
            return NhwcToNchwFn.apply(img, self.vae_model.config.out_channels, z.dtype)
        adt = self.vae_model.compute_dtype
        zc = z.contiguous()
        if zc.dtype != adt:
            zc = zc.to(adt)
        zs = ops.scale_shift(zc, 1.0 / float(self.vae_model.config.scaling_factor), float(self.shift_factor))
        use_tiling = z.shape[2] > self.tiling_size[0] or z.shape[3] > self.tiling_size[1]
        if not use_tiling:
            img = streams.run_halves(self.vae_model.decode_nhwc, ops.nchw_to_nhwc(zs, dtype=adt))
            return ops.nhwc_to_nchw(img, self.vae_model.config.out_channels, out_dtype=z.dtype)
        return self._tiled_decode(zs)

    def _tiled_decode(self, zs: torch.Tensor, tile_batch: int = 16) -> torch.Tensor:
        """Overlapping latent tiles, each zero-padded to the tile size, decoded and merged with the
        reference's Gaussian weights (autoencoderKL.py:92-131, models/utils.py:156-258).  Tiles of the
        whole batch are decoded together; the merge accumulates in f32 on the device."""
        B, C, H, W = zs.shape
        th, tw = self.tiling_size
        f = self.downsampling_factor
        oc = self.vae_model.config.out_channels
        ys = tile_starts(H, th, self.tiling_overlap[0])
        xs = tile_starts(W, tw, self.tiling_overlap[1])
        acc = torch.zeros(B, oc, H * f, W * f, dtype=torch.float32, device=zs.device)
        wsum = torch.zeros(H * f, W * f, dtype=torch.float32, device=zs.device)
        jobs = [(y0, x0) for y0 in ys for x0 in xs]
        per_call = max(1, tile_batch // B)
        for j0 in range(0, len(jobs), per_call):
            group = jobs[j0:j0 + per_call]
            tiles = []
            for (y0, x0) in group:
                t = zs[:, :, y0:y0 + th, x0:x0 + tw]
                t = torch.nn.functional.pad(t, (0, tw - t.shape[3], 0, th - t.shape[2]))   # zero pad to the tile size
                tiles.append(t)
            dec = streams.run_halves(self.vae_model.decode_nhwc, ops.nchw_to_nhwc(torch.cat(tiles, 0), dtype=self.vae_model.compute_dtype))   # [len*B, th*f, tw*f, 8]
            for k, (y0, x0) in enumerate(group):
                hh, ww = min(th, H - y0) * f, min(tw, W - x0) * f
                wy = torch.tensor(gaussian_weights_1d(hh, hh / 2), dtype=torch.float32, device=zs.device)
                wx = torch.tensor(gaussian_weights_1d(ww, (ww - 1) / 2), dtype=torch.float32, device=zs.device)
                ops.tile_accumulate(dec[k * B:(k + 1) * B], acc, wsum, wy, wx, hh, ww, y0 * f, x0 * f, oc)
        return ops.tile_finalize(acc, wsum, out_dtype=torch.float32)
