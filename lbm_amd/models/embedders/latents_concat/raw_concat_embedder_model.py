"""Raw (pixel-space) concat conditioning, no VAE
(src/lbm/models/embedders/latents_concat/raw_concat_embedder_model.py:9-41)."""
from typing import Any, Dict

import torch

from ..base import BaseConditioner
from .latents_concat_embedder_config import LatentsConcatEmbedderConfig


class RawConcatEmbedder(BaseConditioner):
    def __init__(self, config: LatentsConcatEmbedderConfig):
        BaseConditioner.__init__(self, config)

    def forward(self, batch: Dict[str, Any], vae=None, *args, **kwargs) -> dict:
        dims_list = [batch[k].shape[-2:] for k in self.config.image_keys] + \
                    [batch[k].shape[-2:] for k in self.config.mask_keys]
        assert all(d == dims_list[0] for d in dims_list), "All images and masks must have the same dimensions."
        outputs = [batch[k] for k in self.config.mask_keys] + [batch[k] for k in self.config.image_keys]
        outputs = torch.concat(outputs, dim=1)
        return self.by_rank(outputs)
