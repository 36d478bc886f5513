from .lbm_config import LBMConfig
from .lbm_model import LBMModel

__all__ = ["LBMModel", "LBMConfig"]
