"""Denoiser wrapper with the reference's call protocol
(src/lbm/models/unets/unet.py:55-148): ``denoiser(sample=, timestep=, conditioning=)``."""
from typing import Dict, List, Optional, Union

import torch

from .unet2d_condition import UNet2DConditionModel


class DiffusersUNet2DCondWrapper(UNet2DConditionModel):
    def __init__(self, *args, **kwargs):
        UNet2DConditionModel.__init__(self, *args, **kwargs)

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                conditioning: Dict[str, torch.Tensor], ip_adapter_cond_embedding: Optional[List[torch.Tensor]] = None,
                down_block_additional_residuals: torch.Tensor = None, mid_block_additional_residual: torch.Tensor = None,
                down_intrablock_additional_residuals: torch.Tensor = None, *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"
        class_labels = conditioning["cond"].get("vector", None)
        crossattn = conditioning["cond"].get("crossattn", None)
        concat = conditioning["cond"].get("concat", None)
        if concat is not None:
            sample = torch.cat([sample, concat.to(sample.dtype)], dim=1)
        if ip_adapter_cond_embedding is not None:
            raise NotImplementedError("IP-adapter embeddings are not used by any of the reference's builders")
        return UNet2DConditionModel.forward(
            self, sample=sample, timestep=timestep, encoder_hidden_states=crossattn, class_labels=class_labels,
            down_block_additional_residuals=down_block_additional_residuals,
            mid_block_additional_residual=mid_block_additional_residual,
            down_intrablock_additional_residuals=down_intrablock_additional_residuals).sample

    def freeze(self):
        self.eval()
        for param in self.parameters():
            param.requires_grad = False
