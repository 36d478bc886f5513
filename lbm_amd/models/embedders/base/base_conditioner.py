from typing import Any, Dict

from ...base.base_model import BaseModel
from .base_conditioner_config import BaseConditionerConfig

DIM2CONDITIONING = {2: "vector", 3: "crossattn", 4: "concat"}


class BaseConditioner(BaseModel):
    """Same contract as src/lbm/models/embedders/base/base_conditioner.py of the reference."""

    def __init__(self, config: BaseConditionerConfig):
        BaseModel.__init__(self, config)
        self.config = config
        self.input_key = config.input_key
        self.dim2outputkey = DIM2CONDITIONING
        self.ucg_rate = config.unconditional_conditioning_rate

    def forward(self, batch: Dict[str, Any], force_zero_embedding: bool = False, *args, **kwargs):
        raise NotImplementedError("Forward pass must be implemented in child class")
