from .latents_concat_embedder_config import LatentsConcatEmbedderConfig
from .latents_concat_embedder_model import LatentsConcatEmbedder
from .raw_concat_embedder_model import RawConcatEmbedder

__all__ = ["LatentsConcatEmbedder", "LatentsConcatEmbedderConfig", "RawConcatEmbedder"]
