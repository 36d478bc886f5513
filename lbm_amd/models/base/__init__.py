from .base_model import BaseModel
from .model_config import ModelConfig

__all__ = ["BaseModel", "ModelConfig"]
