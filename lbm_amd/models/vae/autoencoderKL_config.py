from typing import Tuple

from pydantic.dataclasses import dataclass

from ..base import ModelConfig


@dataclass
class AutoencoderKLDiffusersConfig(ModelConfig):
    """Same fields as the reference (src/lbm/models/vae/autoencoderKL_config.py:8-27).
    ``version`` may be a local directory in the diffusers layout."""

    version: str = "stabilityai/sdxl-vae"
    subfolder: str = ""
    revision: str = "main"
    input_key: str = "image"
    tiling_size: Tuple[int, int] = (64, 64)
    tiling_overlap: Tuple[int, int] = (16, 16)
