"""Mirror of the reference's ``BaseModel`` (src/lbm/models/base/base_model.py:9-66):
an nn.Module that tracks ``.device`` / ``.dtype`` through ``.to()``."""
from typing import Any, Dict

import torch
import torch.nn as nn

from .model_config import ModelConfig


class BaseModel(nn.Module):
    def __init__(self, config: ModelConfig):
        nn.Module.__init__(self)
        self.config = config
        self.input_key = config.input_key
        self.device = torch.device("cpu")
        self.dtype = torch.float32

    def on_fit_start(self, device: torch.device | None = None, *args, **kwargs):
        if device is not None:
            self.device = device
        self.to(self.device)

    def forward(self, batch: Dict[str, Any], *args, **kwargs):
        raise NotImplementedError("forward method is not implemented")

    def freeze(self):
        self.eval()
        for param in self.parameters():
            param.requires_grad = False

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        self = super().to(device=device, dtype=dtype, non_blocking=non_blocking)
        if device is not None:
            self.device = device
        if dtype is not None:
            self.dtype = dtype
        return self

    def compute_metrics(self, batch: Dict[str, Any], *args, **kwargs):
        return {}

    def sample(self, batch: Dict[str, Any], *args, **kwargs):
        return {}

    def log_samples(self, batch: Dict[str, Any], *args, **kwargs):
        return None

    def on_train_batch_end(self, batch: Dict[str, Any], *args, **kwargs):
        pass
