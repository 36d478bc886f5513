"""Tiling helpers with the semantics of the reference's ``Tiler`` / ``pad``
(src/lbm/models/utils.py:12-258, :333-349).

The tile grid and the Gaussian weights are host logic and live here; the weighted
accumulation itself runs in the ``lbmk_tile_accumulate`` / ``lbmk_tile_finalize`` kernels
(see ``AutoencoderKLDiffusers._tiled_decode``).  ``Tiler`` keeps the reference's object API
for callers that tile something else than the VAE decode.
"""
import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

TILING_METHODS = ["average", "gaussian"]


def tile_starts(size: int, tile: int, overlap: int) -> List[int]:
    """Tile origins along one axis: range(0, size, tile - overlap); no overlap when the input
    fits in one tile (utils.py:42-46, 69-71)."""
    ov = overlap if size > tile else 0
    return list(range(0, size, tile - ov))


def gaussian_weights_1d(n: int, midpoint: float) -> np.ndarray:
    """exp(-(x-mid)^2 / n^2 / (2*0.01)) / sqrt(2*pi*0.01) for x in 0..n-1 (utils.py:175-199).
    The reference uses midpoint (n-1)/2 along x and n/2 along y; callers pass it explicitly."""
    var = 0.01
    x = np.arange(n, dtype=np.float64)
    return np.exp(-(x - midpoint) * (x - midpoint) / (n * n) / (2 * var)) / math.sqrt(2 * math.pi * var)


def pad(x: torch.Tensor, base_h: int, base_w: int) -> torch.Tensor:
    """Zero-pads bottom / right up to the next multiple of (base_h, base_w)."""
    h, w = x.shape[-2:]
    h_ = math.ceil(h / base_h) * base_h
    w_ = math.ceil(w / base_w) * base_w
    if w_ != w or h_ != h:
        x = F.pad(x, (0, w_ - w, 0, h_ - h))
    return x


class Tiler:
    def get_tiles(self, input: torch.Tensor, tile_size: tuple, overlap_size: tuple, scale: int = 1,
                  out_channels: int = 3) -> List[List[torch.Tensor]]:
        assert overlap_size[0] <= tile_size[0] and overlap_size[1] <= tile_size[1], \
            f"Overlap size {overlap_size} must be smaller than tile size {tile_size}"
        B, C, H, W = input.shape
        th, tw = tile_size
        self._ys = tile_starts(H, th, overlap_size[0])
        self._xs = tile_starts(W, tw, overlap_size[1])
        self.output_overlap_size = (int((overlap_size[0] if H > th else 0) * scale),
                                    int((overlap_size[1] if W > tw else 0) * scale))
        self.tile_size = tile_size
        self.output_tile_size = (int(th * scale), int(tw * scale))
        self.output_shape = (B, out_channels, int(H * scale), int(W * scale))
        self._scale = scale
        return [[input[:, :, i:i + th, j:j + tw].clone() for j in self._xs] for i in self._ys]

    def _gaussian_weights(self, tile_width: int, tile_height: int, nbatches: int, channels: int) -> torch.Tensor:
        w = np.outer(gaussian_weights_1d(tile_height, tile_height / 2), gaussian_weights_1d(tile_width, (tile_width - 1) / 2))
        return torch.tile(torch.tensor(w), (nbatches, channels, 1, 1))

    def merge_tiles(self, tiles: List[List[torch.Tensor]], tiling_method: str = "gaussian") -> torch.Tensor:
        if tiling_method not in TILING_METHODS:
            raise ValueError(f"Unknown tiling method {tiling_method}. Available methods are {TILING_METHODS}")
        B, C, OH, OW = self.output_shape
        dev = tiles[0][0].device
        out = torch.zeros(self.output_shape, device=dev)
        wsum = torch.zeros(self.output_shape, device=dev)
        for a, y0 in enumerate(self._ys):
            for b, x0 in enumerate(self._xs):
                t = tiles[a][b]
                if tiling_method == "gaussian":
                    w = self._gaussian_weights(t.shape[3], t.shape[2], B, C).to(device=dev, dtype=torch.float32)
                else:
                    w = torch.ones_like(t, dtype=torch.float32)
                oy, ox = int(y0 * self._scale), int(x0 * self._scale)
                out[:, :, oy:oy + t.shape[2], ox:ox + t.shape[3]] += t.float() * w
                wsum[:, :, oy:oy + t.shape[2], ox:ox + t.shape[3]] += w
        return out / wsum


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


@torch.no_grad()
def update_ema(target_params, source_params, rate: float = 0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)
