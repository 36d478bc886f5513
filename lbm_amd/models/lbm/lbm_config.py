from typing import List, Literal, Optional

from pydantic.dataclasses import dataclass

from ..base import ModelConfig


@dataclass
class LBMConfig(ModelConfig):
    """Same fields, defaults and validation as the reference's LBMConfig
    (src/lbm/models/lbm/lbm_config.py:70-103)."""

    source_key: str = "source_image"
    target_key: str = "target_image"
    mask_key: Optional[str] = None
    latent_loss_weight: float = 1.0
    latent_loss_type: Literal["l2", "l1"] = "l2"
    pixel_loss_type: Literal["l2", "l1", "lpips"] = "l2"
    pixel_loss_max_size: int = 512
    pixel_loss_weight: float = 0.0
    timestep_sampling: Literal["uniform", "log_normal", "custom_timesteps"] = "uniform"
    logit_mean: Optional[float] = 0.0
    logit_std: Optional[float] = 1.0
    selected_timesteps: Optional[List[float]] = None
    prob: Optional[List[float]] = None
    bridge_noise_sigma: float = 0.001
    lighting_condition_weight: float = 1.0
    concat_condition_weight: float = 1.0

    def __post_init__(self):
        super().__post_init__()
        if self.timestep_sampling == "log_normal":
            assert isinstance(self.logit_mean, float) and isinstance(self.logit_std, float), \
                "logit_mean and logit_std should be float for log_normal timestep sampling"
        if self.timestep_sampling == "custom_timesteps":
            assert isinstance(self.selected_timesteps, list) and isinstance(self.prob, list), \
                "timesteps and prob should be list for custom_timesteps timestep sampling"
            assert len(self.selected_timesteps) == len(self.prob), \
                "timesteps and prob should be of same length for custom_timesteps timestep sampling"
            assert sum(self.prob) == 1, "prob should sum to 1 for custom_timesteps timestep sampling"
