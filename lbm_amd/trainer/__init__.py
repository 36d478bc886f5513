from .training_config import TrainingConfig
from .utils import StateDictAdapter, StateDictRenamer

__all__ = ["TrainingConfig", "StateDictAdapter", "StateDictRenamer"]
