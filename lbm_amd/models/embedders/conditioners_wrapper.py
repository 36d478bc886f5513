"""ConditionerWrapper with the reference's merge rules
(src/lbm/models/embedders/conditioners_wrapper.py:9-114)."""
from typing import Any, Dict, List, Union

import torch
import torch.nn as nn

from .base import BaseConditioner

KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}


class ConditionerWrapper(nn.Module):
    def __init__(self, conditioners: Union[List[BaseConditioner], None] = None):
        nn.Module.__init__(self)
        self.conditioners = nn.ModuleList(conditioners)
        self.device = torch.device("cpu")
        self.dtype = torch.float32

    def on_fit_start(self, device: torch.device | None = None, *args, **kwargs):
        for conditioner in self.conditioners:
            conditioner.on_fit_start(device=device, *args, **kwargs)

    def forward(self, batch: Dict[str, Any], ucg_keys: List[str] = None, set_ucg_rate_zero=False, *args, **kwargs):
        if ucg_keys is None:
            ucg_keys = []
        wrapper_outputs = dict(cond={})
        for conditioner in self.conditioners:
            if conditioner.input_key in ucg_keys:
                force_zero_embedding = True
            elif conditioner.ucg_rate > 0 and not set_ucg_rate_zero:
                force_zero_embedding = bool(torch.rand(1) < conditioner.ucg_rate)
            else:
                force_zero_embedding = False
            conditioner_output = conditioner.forward(batch, force_zero_embedding=force_zero_embedding, *args, **kwargs)
            for key in conditioner_output:
                if key in wrapper_outputs["cond"]:
                    wrapper_outputs["cond"][key] = torch.cat(
                        [wrapper_outputs["cond"][key], conditioner_output[key]], KEY2CATDIM[key])
                else:
                    wrapper_outputs["cond"][key] = conditioner_output[key]
        return wrapper_outputs

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        self = super().to(device=device, dtype=dtype, non_blocking=non_blocking)
        for conditioner in self.conditioners:
            conditioner.to(device=device, dtype=dtype, non_blocking=non_blocking)
        if device is not None:
            self.device = device
        if dtype is not None:
            self.dtype = dtype
        return self
