"""``evaluate``: one PIL image through encode -> bridge sampling -> decode, with the
reference's pre/post-processing (src/lbm/inference/inference.py:12-70): aspect-ratio
bucket, ``*2-1`` scaling, clamp, ``(x+1)/2``, and — like the reference, whose final
``resize`` result is discarded (:68) — no resize back to the input size."""
import logging

import numpy as np
import PIL.Image
import torch

from ..models.lbm import LBMModel

logger = logging.getLogger(__name__)

ASPECT_RATIOS = {
    str(512 / 2048): (512, 2048),
    str(1024 / 1024): (1024, 1024),
    str(2048 / 512): (2048, 512),
    str(896 / 1152): (896, 1152),
    str(1152 / 896): (1152, 896),
    str(512 / 1920): (512, 1920),
    str(640 / 1536): (640, 1536),
    str(768 / 1280): (768, 1280),
    str(1280 / 768): (1280, 768),
    str(1536 / 640): (1536, 640),
    str(1920 / 512): (1920, 512),
}


def _to_tensor(img: PIL.Image.Image) -> torch.Tensor:
    """torchvision ToTensor for 8-bit images: HWC uint8 -> CHW float32 in [0, 1]."""
    arr = np.asarray(img.convert("RGB") if img.mode not in ("RGB", "L") else img, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)


def _to_pil(x: torch.Tensor) -> PIL.Image.Image:
    """torchvision ToPILImage for float CHW tensors: mul(255) then truncating cast to uint8."""
    arr = x.mul(255).byte().permute(1, 2, 0).numpy()
    return PIL.Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr)


def closest_bucket(source_image: PIL.Image.Image):
    ori_h_bg, ori_w_bg = source_image.size      # (PIL size is (width, height); the reference names them h, w)
    ar_bg = ori_h_bg / ori_w_bg
    closest_ar_bg = min(ASPECT_RATIOS, key=lambda x: abs(float(x) - ar_bg))
    return ASPECT_RATIOS[closest_ar_bg]


@torch.no_grad()
def evaluate(model: LBMModel, source_image: PIL.Image.Image, num_sampling_steps: int = 1):
    source_dimensions = closest_bucket(source_image)
    source_image = source_image.resize(source_dimensions)
    img_pasted_tensor = _to_tensor(source_image).unsqueeze(0) * 2 - 1
    batch = {"source_image": img_pasted_tensor.cuda().to(torch.bfloat16)}
    z_source = model.vae.encode(batch[model.source_key])
    output_image = model.sample(z=z_source, num_steps=num_sampling_steps, conditioner_inputs=batch,
                                max_samples=1).clamp(-1, 1)
    output_image = (output_image[0].float().cpu() + 1) / 2
    return _to_pil(output_image)
