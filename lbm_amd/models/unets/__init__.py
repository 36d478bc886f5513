from .unet import DiffusersUNet2DCondWrapper
from .unet2d import DiffusersUNet2DWrapper, UNet2DModel
from .unet2d_condition import UNet2DConditionModel

__all__ = ["DiffusersUNet2DCondWrapper", "DiffusersUNet2DWrapper", "UNet2DConditionModel", "UNet2DModel"]
