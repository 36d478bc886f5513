from dataclasses import field
from typing import List, Literal, Optional, Union

from pydantic.dataclasses import dataclass

from ..config import BaseConfig


@dataclass
class TrainingConfig(BaseConfig):
    """Same fields and defaults as the reference's TrainingConfig (src/lbm/trainer/training_config.py:52-82)."""

    experiment_id: Optional[str] = None
    optimizer_name: Literal["Adam", "AdamW", "Adadelta", "Adagrad", "RMSprop", "SGD"] = field(default_factory=lambda: "AdamW")
    optimizer_kwargs: Optional[dict] = field(default_factory=lambda: {})
    learning_rate: float = field(default_factory=lambda: 1e-3)
    lr_scheduler_name: Optional[Literal["StepLR", "CosineAnnealingLR", "CosineAnnealingWarmRestarts",
                                        "ReduceLROnPlateau", "ExponentialLR", None]] = None
    lr_scheduler_kwargs: Optional[dict] = field(default_factory=lambda: {})
    lr_scheduler_interval: Optional[Literal["step", "epoch", None]] = "step"
    lr_scheduler_frequency: Optional[int] = 1
    metrics: Optional[List[str]] = None
    tracking_metrics: Optional[List[str]] = None
    backup_every: int = 50
    trainable_params: List[str] = field(default_factory=lambda: ["./*"])
    log_keys: Optional[Union[str, List[str]]] = "txt"
    log_samples_model_kwargs: Optional[dict] = field(
        default_factory=lambda: {"max_samples": 4, "num_steps": 20, "input_shape": None})
