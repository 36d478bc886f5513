"""Base class of the conditioners (contract: src/lbm/models/embedders/base/base_conditioner.py:8-60).

A conditioner maps a batch dict to ``{kind: tensor}``, the kind fixed by the tensor's rank: 2-D "vector",
3-D "crossattn" (tokens), 4-D "concat" (channels appended to the denoiser input).  ``ucg_rate`` is the
probability with which ConditionerWrapper asks for a dropped ("unconditional") embedding during training.
"""
from typing import Any, Dict

import torch

from ...base.base_model import BaseModel
from .base_conditioner_config import BaseConditionerConfig

DIM2CONDITIONING = {2: "vector", 3: "crossattn", 4: "concat"}


class BaseConditioner(BaseModel):
    dim2outputkey = DIM2CONDITIONING

    def __init__(self, config: BaseConditionerConfig):
        super().__init__(config)
        self.ucg_rate = config.unconditional_conditioning_rate

    def by_rank(self, embedding: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The conditioner's return value for one embedding tensor."""
        return {self.dim2outputkey[embedding.dim()]: embedding}

    def forward(self, batch: Dict[str, Any], force_zero_embedding: bool = False, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement forward()")
