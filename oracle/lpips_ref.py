"""CPU restatement of lpips 0.1.4's LPIPS(net="vgg") forward in plain torch (fp32): test infrastructure only.

The package is a dependency of the reference (requirements.txt; call sites src/lbm/models/lbm/lbm_model.py:83,
248-249), not part of its tree and not installed here, and no weights or golden outputs for it exist on disk:
PARITY UNPINNED.  Restated from the published algorithm: ScalingLayer, torchvision VGG16 features split at
relu1_2 / relu2_2 / relu3_3 / relu4_3 / relu5_3, channel-wise unit normalisation with eps 1e-10, squared
difference, 1x1 linear heads without bias, spatial average, sum over the five layers.
``sd`` uses the package's state-dict key names.
"""
import torch
import torch.nn.functional as F

_CONVS = {1: [0, 2], 2: [5, 7], 3: [10, 12, 14], 4: [17, 19, 21], 5: [24, 26, 28]}


def normalize_tensor(x, eps=1e-10):
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def vgg_taps(sd, x):
    taps = []
    for k in range(1, 6):
        if k > 1:
            x = F.max_pool2d(x, 2, 2)
        for i in _CONVS[k]:
            x = F.relu(F.conv2d(x, sd[f"net.slice{k}.{i}.weight"], sd[f"net.slice{k}.{i}.bias"], padding=1))
        taps.append(x)
    return taps


def lpips_forward(sd, in0, in1, keep=None):
    """-> [B, 1, 1, 1].  ``keep``: optional list of five {0, 1} masks [B, C, H, W] (train-mode dropout, p = 0.5)."""
    shift, scale = sd["scaling_layer.shift"], sd["scaling_layer.scale"]
    f0 = vgg_taps(sd, (in0 - shift) / scale)
    f1 = vgg_taps(sd, (in1 - shift) / scale)
    val = 0
    for k in range(5):
        d = (normalize_tensor(f0[k]) - normalize_tensor(f1[k])) ** 2
        if keep is not None:
            d = d * keep[k] * 2.0
        val = val + F.conv2d(d, sd[f"lin{k}.model.1.weight"]).mean(dim=(2, 3), keepdim=True)
    return val
