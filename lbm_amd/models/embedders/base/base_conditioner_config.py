from pydantic.dataclasses import dataclass

from ....config import BaseConfig


@dataclass
class BaseConditionerConfig(BaseConfig):
    input_key: str = "text"
    unconditional_conditioning_rate: float = 0.0

    def __post_init__(self):
        super().__post_init__()
        assert 0.0 <= self.unconditional_conditioning_rate <= 1.0, \
            "Unconditional conditioning rate should be between 0 and 1"
