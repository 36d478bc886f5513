from .unet import DiffusersUNet2DCondWrapper
from .unet2d_condition import UNet2DConditionModel

__all__ = ["DiffusersUNet2DCondWrapper", "UNet2DConditionModel"]
