from typing import Tuple

from pydantic.dataclasses import dataclass

from ..base import ModelConfig


@dataclass
class AutoencoderKLDiffusersConfig(ModelConfig):
    """The reference's fields (src/lbm/models/vae/autoencoderKL_config.py:8-27) plus ``allow_random_init``.
    ``version`` is a local directory in the diffusers layout; a hub name raises (no network) unless
    ``allow_random_init`` is set, which builds the named backbone's VAE with untrained weights."""

    version: str = "stabilityai/sdxl-vae"
    subfolder: str = ""
    revision: str = "main"
    input_key: str = "image"
    tiling_size: Tuple[int, int] = (64, 64)
    tiling_overlap: Tuple[int, int] = (16, 16)
    allow_random_init: bool = False
