"""LPIPS perceptual distance, VGG16 variant, on the HIP kernels.

The reference builds ``lpips.LPIPS(net="vgg")`` as ``LBMModel.lpips_loss`` when ``pixel_loss_type == "lpips"``
(src/lbm/models/lbm/lbm_model.py:82-83) and evaluates ``lpips_loss(decoded * mask, target * mask).mean()``
(:248-249).  The ``lpips`` package (0.1.4, requirements.txt) is not part of the reference tree; this module
restates its published algorithm with the same module tree, so a checkpoint's ``lpips_loss.*`` keys load with
``strict=True``:

  scaling_layer.shift / .scale                     buffers [1, 3, 1, 1]
  net.slice{1..5}.{torchvision vgg16.features idx}.weight / .bias      13 convolutions
  lin{0..4}.model.1.weight  and the same tensors again as  lins.{0..4}.model.1.weight     [1, C, 1, 1]

distance = sum over the five ReLU taps of  mean_hw( sum_c w_c * (f0_c/(|f0|+1e-10) - f1_c/(|f1|+1e-10))^2 ).
Convolutions run as implicit GEMMs with a fused ReLU; pooling, the input scaling and the per-layer head are
dedicated kernels, each with a hand-written backward (the gradient flows to the first image only; the second
is the target).  As in the package, the 1x1 heads sit behind ``nn.Dropout``: when the module is in train mode
(the reference's trainer puts the whole model in train mode) half of the squared differences are dropped
and the rest doubled, with a mask drawn from torch's device generator.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from ..layers import HipConv2d

# torchvision vgg16.features: index -> (in, out) of the convolutions; 'M' positions are the 2x2 max pools
_VGG = [(0, 3, 64), (2, 64, 64), "M", (5, 64, 128), (7, 128, 128), "M", (10, 128, 256), (12, 256, 256), (14, 256, 256), "M",
        (17, 256, 512), (19, 512, 512), (21, 512, 512), "M", (24, 512, 512), (26, 512, 512), (28, 512, 512)]
_SLICES = {1: range(0, 4), 2: range(4, 9), 3: range(9, 16), 4: range(16, 23), 5: range(23, 30)}
_CHNS = [64, 128, 256, 512, 512]


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])


class _Vgg16Slices(nn.Module):
    """The five feature slices; children are named by their torchvision index (ReLU / pooling hold no state)."""

    def __init__(self):
        super().__init__()
        convs = {e[0]: e for e in _VGG if e != "M"}
        for k, idxs in _SLICES.items():
            seq = nn.Sequential()
            for i in idxs:
                if i in convs:
                    seq.add_module(str(i), HipConv2d(convs[i][1], convs[i][2], 3, padding=1))
                elif i in (4, 9, 16, 23):
                    seq.add_module(str(i), nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    seq.add_module(str(i), nn.ReLU(inplace=True))
            setattr(self, f"slice{k}", seq)
        for p in self.parameters():
            p.requires_grad = False

    def features(self, x):
        """x: NHWC bf16 [B, H, W, 8] (scaled image) -> the five ReLU taps, NHWC bf16"""
        from ...autograd import MaxPool2Fn
        taps = []
        for k in range(1, 6):
            for m in getattr(self, f"slice{k}"):
                if isinstance(m, HipConv2d):
                    x = m(x, act=ops.ACT_RELU)
                elif isinstance(m, nn.MaxPool2d):
                    x = MaxPool2Fn.apply(x) if (torch.is_grad_enabled() and x.requires_grad) else ops.maxpool2(x)
            taps.append(x)
        return taps


class NetLinLayer(nn.Module):
    def __init__(self, chn_in: int, chn_out: int = 1, use_dropout: bool = True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class LPIPS(nn.Module):
    def __init__(self, net: str = "vgg", use_dropout: bool = True, eval_mode: bool = True):
        super().__init__()
        if net not in ("vgg", "vgg16"):
            raise NotImplementedError(f"LPIPS backbone {net!r}: only the VGG16 variant the reference uses is built")
        self.pnet_type, self.chns, self.L = net, _CHNS, len(_CHNS)
        self.scaling_layer = ScalingLayer()
        self.net = _Vgg16Slices()
        for i, c in enumerate(_CHNS):
            setattr(self, f"lin{i}", NetLinLayer(c, use_dropout=use_dropout))
        self.lins = nn.ModuleList([getattr(self, f"lin{i}") for i in range(self.L)])
        if eval_mode:
            self.eval()

    def _prep(self, x, mask, clamp):
        from ...autograd import LpipsPrepFn
        shift = self.scaling_layer.shift.flatten().tolist()
        scale = self.scaling_layer.scale.flatten().tolist()
        if x.dtype not in (torch.bfloat16, torch.float32):
            x = x.float()
        if torch.is_grad_enabled() and x.requires_grad:
            return LpipsPrepFn.apply(x, mask, shift, scale, clamp)
        return ops.lpips_prep(x.contiguous(), mask, shift, scale, clamp)

    def forward(self, in0: torch.Tensor, in1: torch.Tensor, normalize: bool = False, *, mask: Optional[torch.Tensor] = None,
                clamp0: bool = False) -> torch.Tensor:
        """in0, in1: NCHW RGB in [-1, 1] ([0, 1] with ``normalize``) -> [B, 1, 1, 1].  ``mask`` ([B, 1, H, W]) multiplies
        both images and ``clamp0`` clips in0 to [-1, 1] first: the reference's ``lpips_loss(clamp(dec) * mask, target * mask)``
        without extra passes over the images."""
        from ...autograd import LpipsHeadFn
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        with torch.no_grad():
            f1 = self.net.features(self._prep(in1.detach(), mask, False))
        f0 = self.net.features(self._prep(in0, mask, clamp0))
        total = None
        for k in range(self.L):
            lin = self.lins[k].model
            w = lin[-1].weight.detach().float().reshape(-1).contiguous()
            keep = None
            if isinstance(lin[0], nn.Dropout) and lin[0].training:
                B, H, W, Cc = f0[k].shape
                keep = (torch.rand(B, Cc, H, W, device=f0[k].device) >= lin[0].p).permute(0, 2, 3, 1).contiguous().to(torch.uint8)
            if torch.is_grad_enabled() and f0[k].requires_grad:
                d = LpipsHeadFn.apply(f0[k], f1[k], w, keep)
            else:
                d = ops.lpips_head(f0[k], f1[k], w, keep)
            total = d if total is None else total + d
        return total.view(-1, 1, 1, 1)
