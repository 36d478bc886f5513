from .autoencoder_kl import AutoencoderKL
from .autoencoderKL import AutoencoderKLDiffusers
from .autoencoderKL_config import AutoencoderKLDiffusersConfig

__all__ = ["AutoencoderKLDiffusers", "AutoencoderKLDiffusersConfig", "AutoencoderKL"]
