"""Lighting-parameter conditioning for the fill-light model: a [B, P] vector of scene parameters becomes
``len(param_groups)`` cross-attention tokens of width ``out_dim``.

Contract (src/lbm/models/embedders/lighting_params_embedder.py:8-88): token i = MLP_i(Fourier_i(params[:, group_i]))
with Fourier(x) = [sin(2 pi x B^T), cos(2 pi x B^T)], B a fixed Gaussian buffer of scale ``scale``; the MLP is
Linear-SiLU-Linear-SiLU-Linear.  State-dict names ``group_fourier.<i>.B`` and ``group_mlps.<i>.{0,2,4}.{weight,bias}``.
Every forward consumes one ``torch.rand(1)`` from the CPU generator after the tokens are computed (the reference
spends it on a 1-in-100 statistics print, :83); the draw is kept so that the random stream of a training step — ucg
decisions, timestep, bridge noise — stays where the reference has it.  A few kFLOP of glue: plain PyTorch on the
model's device, no kernel.
"""
import math
from typing import List, Optional

import torch
from pydantic.dataclasses import dataclass
from torch import nn

from .base import BaseConditioner, BaseConditionerConfig


@dataclass
class LightingParamsEmbedderConfig(BaseConditionerConfig):
    mapping_size: int = 256
    scale: float = 10.0
    hidden_dim: int = 512
    out_dim: int = 768
    seq_len: int = 4
    param_groups: Optional[List[List[int]]] = None
    input_key: str = "lighting_params"


class GaussianFourierFeatures(nn.Module):
    """x [B, in_features] -> [B, 2 * mapping_size]: sines then cosines of random projections."""

    def __init__(self, in_features: int, mapping_size: int = 256, scale: float = 10.0):
        super().__init__()
        self.register_buffer("B", scale * torch.randn(mapping_size, in_features))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        angles = (2 * math.pi * x.to(self.B.dtype)) @ self.B.T
        return torch.cat((angles.sin(), angles.cos()), dim=-1)


def _token_mlp(width_in: int, hidden: int, width_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(width_in, hidden), nn.SiLU(), nn.Linear(hidden, hidden), nn.SiLU(),
                         nn.Linear(hidden, width_out))


class LightingParamsEmbedder(BaseConditioner):
    def __init__(self, config: LightingParamsEmbedderConfig):
        super().__init__(config)
        groups = config.param_groups
        if groups is None or len(groups) != config.seq_len:
            raise ValueError(f"seq_len ({config.seq_len}) must equal the number of param_groups "
                             f"({'none given' if groups is None else len(groups)})")
        self.param_groups = groups
        self.seq_len, self.out_dim = config.seq_len, config.out_dim
        # registration order fixes the order of the construction-time random draws: all Fourier buffers, then the MLPs
        self.group_fourier = nn.ModuleList(GaussianFourierFeatures(len(g), config.mapping_size, config.scale)
                                           for g in groups)
        self.group_mlps = nn.ModuleList(_token_mlp(2 * config.mapping_size, config.hidden_dim, config.out_dim)
                                        for _ in groups)

    def token(self, i: int, params: torch.Tensor) -> torch.Tensor:
        return self.group_mlps[i](self.group_fourier[i](params[:, self.param_groups[i]]))

    def forward(self, batch, force_zero_embedding: bool = False, *args, **kwargs):
        params = batch[self.input_key]
        if params.dim() == 1:
            params = params[None]
        tokens = torch.stack([self.token(i, params) for i in range(self.seq_len)], dim=1)
        torch.rand(1)      # see the module docstring: part of the reproducible CPU random stream
        return self.by_rank(tokens)
