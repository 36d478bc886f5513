"""``BaseModel``: the nn.Module base of LBMModel, the VAE wrapper and every conditioner.

Contract taken from the reference (src/lbm/models/base/base_model.py:9-66): the module carries
``config`` / ``input_key`` and public ``device`` / ``dtype`` attributes that follow ``.to(...)``,
``on_fit_start(device)`` moves it, ``freeze()`` puts it in eval mode with no trainable parameter, and the
training hooks (``compute_metrics``, ``sample``, ``log_samples``, ``on_train_batch_end``) default to no-ops.
"""
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .model_config import ModelConfig


def placement_of(args: tuple, kwargs: dict) -> Tuple[Optional[torch.device], Optional[torch.dtype]]:
    """(device, dtype) named by the arguments of an ``nn.Module.to`` call, ``None`` where unspecified.
    Accepts what ``Module.to`` accepts: a device (object, string or index), a dtype, or a tensor to imitate."""
    device, dtype = kwargs.get("device"), kwargs.get("dtype")
    for a in args:
        if isinstance(a, torch.Tensor):
            device, dtype = a.device, a.dtype
        elif isinstance(a, torch.dtype):
            dtype = a
        elif isinstance(a, (torch.device, str, int)):
            device = a
    if isinstance(kwargs.get("tensor"), torch.Tensor):
        device, dtype = kwargs["tensor"].device, kwargs["tensor"].dtype
    if device is not None and not isinstance(device, torch.device):
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    return device, dtype


class PlacementTracking:
    """Mixin for nn.Modules whose ``device`` / ``dtype`` attributes mirror the last ``.to(...)`` (modules
    without parameters, such as the concat conditioners, have nothing else to read them from)."""

    def _init_placement(self):
        self.device = torch.device("cpu")
        self.dtype = torch.float32

    def to(self, *args, **kwargs):
        moved = super().to(*args, **kwargs)
        device, dtype = placement_of(args, kwargs)
        if device is not None:
            moved.device = device
        if dtype is not None:
            moved.dtype = dtype
        return moved


class BaseModel(PlacementTracking, nn.Module):
    def __init__(self, config: ModelConfig):
        nn.Module.__init__(self)
        self._init_placement()
        self.config = config
        self.input_key = config.input_key

    def on_fit_start(self, device: Optional[torch.device] = None, *args, **kwargs):
        self.to(self.device if device is None else device)

    def freeze(self):
        self.eval()
        self.requires_grad_(False)

    def forward(self, batch: Dict[str, Any], *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement forward()")

    # hooks a trainer may call; models override what they support
    def compute_metrics(self, batch: Dict[str, Any], *args, **kwargs):
        return {}

    def sample(self, batch: Dict[str, Any], *args, **kwargs):
        return {}

    def log_samples(self, batch: Dict[str, Any], *args, **kwargs):
        return None

    def on_train_batch_end(self, batch: Dict[str, Any], *args, **kwargs):
        return None
