from .conditioners_wrapper import ConditionerWrapper
from .latents_concat import LatentsConcatEmbedder, LatentsConcatEmbedderConfig, RawConcatEmbedder
from .lighting_params_embedder import LightingParamsEmbedder, LightingParamsEmbedderConfig

__all__ = ["LatentsConcatEmbedder", "LatentsConcatEmbedderConfig", "RawConcatEmbedder", "LightingParamsEmbedder",
           "LightingParamsEmbedderConfig", "ConditionerWrapper"]
