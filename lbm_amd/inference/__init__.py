from .inference import evaluate
from .relight import build_filllight_model, build_filllight_refine_model, build_relight_model
from .utils import get_model

__all__ = ["evaluate", "get_model", "build_relight_model", "build_filllight_model", "build_filllight_refine_model"]
