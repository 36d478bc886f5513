from .inference import evaluate
from .utils import get_model

__all__ = ["evaluate", "get_model"]
