from .base_conditioner import BaseConditioner
from .base_conditioner_config import BaseConditionerConfig

__all__ = ["BaseConditioner", "BaseConditionerConfig"]
