"""Lighting-parameter conditioning: per-group Gaussian Fourier features + a 3-layer SiLU MLP
-> cross-attention tokens [B, n_groups, out_dim]
(src/lbm/models/embedders/lighting_params_embedder.py:8-88 of the reference).

This is conditioning glue on a [B, <=7] input (a few kFLOP); it stays plain PyTorch on the
device.  Buffer / parameter names match the reference (``group_fourier.i.B``,
``group_mlps.i.{0,2,4}.{weight,bias}``).  The reference's 1-in-100 debug print is kept out, but
its ``torch.rand(1)`` draw is preserved so the CPU RNG stream stays aligned (:83).
"""
import torch
from pydantic.dataclasses import dataclass
from torch import nn

from .base import BaseConditioner, BaseConditionerConfig


class GaussianFourierFeatures(nn.Module):
    def __init__(self, in_features, mapping_size=256, scale=10.0):
        super().__init__()
        self.register_buffer("B", torch.randn(mapping_size, in_features) * scale)

    def forward(self, x):
        x = x.to(self.B.dtype)      # the reference runs this product under autocast; without it dtypes must agree
        x_proj = 2 * torch.pi * x @ self.B.T
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


@dataclass
class LightingParamsEmbedderConfig(BaseConditionerConfig):
    mapping_size: int = 256
    scale: float = 10.0
    hidden_dim: int = 512
    out_dim: int = 768
    seq_len: int = 4
    param_groups: list[list[int]] | None = None
    input_key: str = "lighting_params"


class LightingParamsEmbedder(BaseConditioner):
    def __init__(self, config: LightingParamsEmbedderConfig):
        BaseConditioner.__init__(self, config)
        self.param_groups = config.param_groups
        if self.param_groups is None or config.seq_len != len(self.param_groups):
            raise ValueError("lighting_embedder_config.seq_len must match the number of param_groups "
                             f"({0 if self.param_groups is None else len(self.param_groups)}).")
        self.group_fourier = nn.ModuleList(
            [GaussianFourierFeatures(len(g), config.mapping_size, config.scale) for g in self.param_groups])
        self.group_mlps = nn.ModuleList([
            nn.Sequential(nn.Linear(config.mapping_size * 2, config.hidden_dim), nn.SiLU(),
                          nn.Linear(config.hidden_dim, config.hidden_dim), nn.SiLU(),
                          nn.Linear(config.hidden_dim, config.out_dim))
            for _ in self.param_groups])
        self.out_dim = config.out_dim
        self.seq_len = config.seq_len

    def forward(self, batch, force_zero_embedding=False, *args, **kwargs):
        lighting_params = batch[self.input_key]
        if lighting_params.dim() == 1:
            lighting_params = lighting_params.unsqueeze(0)
        embeddings = [mlp(fourier(lighting_params[:, group]))
                      for group, fourier, mlp in zip(self.param_groups, self.group_fourier, self.group_mlps)]
        embedding = torch.stack(embeddings, dim=1)
        torch.rand(1)      # the reference draws here for its debug print; keep the RNG stream aligned
        return {self.dim2outputkey[embedding.dim()]: embedding}
