"""Concat conditioning: VAE latents of images + bilinearly resized masks
(src/lbm/models/embedders/latents_concat/latents_concat_embedder_model.py:10-80)."""
from typing import Any, Dict

import torch

from ..base import BaseConditioner
from .latents_concat_embedder_config import LatentsConcatEmbedderConfig


class LatentsConcatEmbedder(BaseConditioner):
    def __init__(self, config: LatentsConcatEmbedderConfig):
        BaseConditioner.__init__(self, config)

    def forward(self, batch: Dict[str, Any], vae, *args, **kwargs) -> dict:
        dims_list = [batch[k].shape[-2:] for k in self.config.image_keys] + \
                    [batch[k].shape[-2:] for k in self.config.mask_keys]
        assert all(d == dims_list[0] for d in dims_list), "All images and masks must have the same dimensions."
        first = self.config.image_keys[0] if len(self.config.image_keys) > 0 else self.config.mask_keys[0]
        latent_dims = (batch[first].shape[-2] // vae.downsampling_factor, batch[first].shape[-1] // vae.downsampling_factor)
        outputs = []
        for mask_key in self.config.mask_keys:
            # torchvision F.resize(BILINEAR) on tensors = interpolate(bilinear, align_corners=False, antialias=True)
            outputs.append(torch.nn.functional.interpolate(batch[mask_key], size=latent_dims, mode="bilinear",
                                                           align_corners=False, antialias=True))
        for image_key in self.config.image_keys:
            outputs.append(vae.encode(batch[image_key]))
        outputs = torch.concat(outputs, dim=1)
        return self.by_rank(outputs)
