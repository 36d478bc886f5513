from .conditioners_wrapper import ConditionerWrapper
from .latents_concat import LatentsConcatEmbedder, LatentsConcatEmbedderConfig, RawConcatEmbedder

__all__ = ["LatentsConcatEmbedder", "LatentsConcatEmbedderConfig", "RawConcatEmbedder", "ConditionerWrapper"]
