"""AutoencoderKL (the SD / SDXL VAE) on the MI355X kernels.

Parameter tree = diffusers' ``AutoencoderKL`` (encoder / decoder / quant_conv /
post_quant_conv with the deprecated-attention-block key names ``to_q, to_k, to_v,
to_out.0, group_norm``), which is what the reference loads through
``AutoencoderKL.from_pretrained`` at src/lbm/models/vae/autoencoderKL.py:20-24.
Activations are bf16 channels-last; the asymmetric stride-2 downsample, the nearest-x2
upsample and every bias / residual add are folded into implicit-GEMM launches.
"""
import json
import math
import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from ..layers import HipConv2d, HipGroupNorm, HipLinear, _Cached
from ..unets.unet2d_condition import Downsample2D, ResnetBlock2D, Upsample2D

VAE_EPS = 1e-6
_SCORE_BYTES_BUDGET = 2 << 30   # f32 score matrices materialised at once by the mid attention


FLASH_HEAD_DIMS = (8, 16, 32, 40, 64, 80, 160)


# LBM_NO_TWO_PASS_ATTN=1: the unfused path with f32 scores (same-box A/B runs)
_TWO_PASS = os.environ.get("LBM_NO_TWO_PASS_ATTN") is None


class VaeAttention(nn.Module, _Cached):
    """Self-attention over a feature map with residual (diffusers' deprecated attention block: group_norm,
    to_q / to_k / to_v with bias, to_out.0).  The VAE mid block uses one head of C = 512; UNetMidBlock2D of
    the pixel-space UNet uses C / 8 heads of 8."""

    def __init__(self, channels: int, groups: int = 32, heads: int = 1, eps: float = VAE_EPS):
        super().__init__()
        self.channels, self.heads = channels, heads
        self.group_norm = HipGroupNorm(groups, channels, eps=eps)
        self.to_q = HipLinear(channels, channels)
        self.to_k = HipLinear(channels, channels)
        self.to_v = HipLinear(channels, channels)
        self.to_out = nn.ModuleList([HipLinear(channels, channels), nn.Dropout(0.0)])

    def _v_weight(self):
        def build():
            return ops.pack_linear_weight(self.to_v.weight, None).w, self.to_v.bias.detach().float().contiguous()
        return self._cached("v", (self.to_v.weight, self.to_v.bias), build)

    def forward(self, x):
        B, H, W, C = x.shape
        N = H * W
        if N % 8:
            raise ValueError(f"VAE attention needs H*W % 8 == 0 (got {H}x{W})")
        xn, x = self.group_norm(x, passthrough=True)
        xn = xn.view(B, N, C)
        D = C // self.heads
        if torch.is_grad_enabled() and (x.requires_grad or self.to_q.weight.requires_grad):
            from ...autograd import AttentionFn
            o = AttentionFn.apply(self.to_q(xn), self.to_k(xn), self.to_v(xn), self.heads, D, 1.0 / math.sqrt(D))
            return self.to_out[0](o, residual=x.view(B, N, C)).view(B, H, W, C)
        if D in FLASH_HEAD_DIMS or xn.dtype == torch.float32:      # fp32 mode: the unfused f32 attention takes any D
            wv, bv = self._v_weight()
            o = ops.flash_attn(self.to_q(xn), self.to_k(xn), ops.gemm_nt(wv, xn, bias_m=bv), self.heads, D)
            return self.to_out[0](o, residual=x.view(B, N, C)).view(B, H, W, C)
        if self.heads != 1:
            raise NotImplementedError(f"spatial attention with {self.heads} heads of {D}")
        q = self.to_q(xn)
        k = self.to_k(xn)
        wv, bv = self._v_weight()
        vt = ops.gemm_nt(wv, xn, bias_m=bv)                     # [B, C, N]
        o = torch.empty(B, N, C, dtype=torch.bfloat16, device=x.device)
        scale = 1.0 / math.sqrt(C)
        if _TWO_PASS and C % 64 == 0:
            # bf16 probabilities only (2 bytes per score resident): chunks of images under the same budget
            nb = max(1, min(B, _SCORE_BYTES_BUDGET // (N * N * 2)))
            for b0 in range(0, B, nb):
                b1 = min(B, b0 + nb)
                o[b0:b1] = ops.attention_two_pass(q[b0:b1], k[b0:b1], vt[b0:b1], scale)
            return self.to_out[0](o, residual=x.view(B, N, C)).view(B, H, W, C)
        nb = max(1, min(B, _SCORE_BYTES_BUDGET // (N * N * 4)))
        for b0 in range(0, B, nb):
            b1 = min(B, b0 + nb)
            s = ops.gemm_nt(q[b0:b1], k[b0:b1], out_dtype=torch.float32, alpha=scale)   # [nb, N, N] f32
            p = ops.softmax_rows(s)
            del s
            o[b0:b1] = ops.gemm_nt(p, vt[b0:b1])
            del p
        return self.to_out[0](o, residual=x.view(B, N, C)).view(B, H, W, C)


class _Level(nn.Module):
    def __init__(self, resnets, downsamplers=None, upsamplers=None, attentions=None):
        super().__init__()
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        if downsamplers is not None:
            self.downsamplers = nn.ModuleList(downsamplers)
        if upsamplers is not None:
            self.upsamplers = nn.ModuleList(upsamplers)


def _mid(c, groups):
    return _Level([ResnetBlock2D(c, c, None, groups, VAE_EPS) for _ in range(2)], attentions=[VaeAttention(c, groups)])


def _run_mid(mid, h):
    h = mid.resnets[0](h)
    h = mid.attentions[0](h)
    return mid.resnets[1](h)


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups, double_z=True):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = HipConv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = ch[0]
        for i, c in enumerate(ch):
            in_c, out_c = out_c, c
            resnets = [ResnetBlock2D(in_c if j == 0 else out_c, out_c, None, groups, VAE_EPS) for j in range(layers_per_block)]
            last = i == len(ch) - 1
            self.down_blocks.append(_Level(resnets, downsamplers=None if last else [Downsample2D(out_c, padding=0)]))
        self.mid_block = _mid(ch[-1], groups)
        self.conv_norm_out = HipGroupNorm(groups, ch[-1], eps=VAE_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = HipConv2d(ch[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                h = r(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0](h)
        h = _run_mid(self.mid_block, h)
        return self.conv_out(self.conv_norm_out(h, silu=True))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = HipConv2d(in_channels, ch[-1], 3, padding=1)
        self.mid_block = _mid(ch[-1], groups)
        self.up_blocks = nn.ModuleList()
        rch = ch[::-1]
        out_c = rch[0]
        for i, c in enumerate(rch):
            prev, out_c = out_c, c
            resnets = [ResnetBlock2D(prev if j == 0 else out_c, out_c, None, groups, VAE_EPS)
                       for j in range(layers_per_block + 1)]
            last = i == len(ch) - 1
            self.up_blocks.append(_Level(resnets, upsamplers=None if last else [Upsample2D(out_c)]))
        self.conv_norm_out = HipGroupNorm(groups, ch[0], eps=VAE_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = HipConv2d(ch[0], out_channels, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = _run_mid(self.mid_block, h)
        for blk in self.up_blocks:
            for r in blk.resnets:
                h = r(h)
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0](h)
        return self.conv_out(self.conv_norm_out(h, silu=True))


class DiagonalGaussianDistribution:
    """Posterior over latents. ``moments``: bf16 NHWC [B, h, w, 2*Cz] (mean | logvar)."""

    def __init__(self, moments: torch.Tensor, latent_channels: int, dtype: torch.dtype):
        self.moments, self.cz, self.dtype = moments, latent_channels, dtype

    def sample(self, generator: Optional[torch.Generator] = None, eps: Optional[torch.Tensor] = None,
               shift: float = 0.0, scale: float = 1.0) -> torch.Tensor:
        """mean + exp(0.5*clamp(logvar,-30,20)) * eps, then (z - shift)*scale; NCHW.
        ``eps`` (NCHW) may be passed explicitly; otherwise it is drawn on the device exactly as
        diffusers does (randn of the latent shape in the model dtype)."""
        B, h, w, _ = self.moments.shape
        if eps is None:
            eps = torch.randn(B, self.cz, h, w, generator=generator, device=self.moments.device, dtype=self.dtype)
        eps = eps.to(device=self.moments.device, dtype=self.dtype).contiguous()
        return ops.posterior_sample(self.moments, eps, self.cz, shift, scale)

    def mode(self) -> torch.Tensor:
        return ops.nhwc_to_nchw(self.moments, self.cz, out_dtype=self.dtype)


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


# architecture shared by the SD1.5 and SDXL VAEs; they differ in weights and scaling_factor
_KNOWN_VAE_CONFIGS = {
    "sd": dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=[128, 256, 512, 512],
               layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215),
    "sdxl": dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=[128, 256, 512, 512],
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025),
}


def known_vae_config(name: str) -> dict:
    """Architecture + scaling factor of the VAE that ships with a known backbone name."""
    low = name.lower()
    if "xl" in low:
        return dict(_KNOWN_VAE_CONFIGS["sdxl"])
    if "stable-diffusion" in low or low in ("sd", "sd15", "offline", "synthetic"):
        return dict(_KNOWN_VAE_CONFIGS["sd"])
    raise ValueError(f"'{name}' names no known backbone (Stable Diffusion 1.x / SDXL): its VAE configuration is unknown "
                     "offline; pass a local diffusers-layout directory")


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, shift_factor=None,
                 latents_mean=None, latents_std=None, **unused):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=list(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
                                      shift_factor=shift_factor, latents_mean=latents_mean, latents_std=latents_std)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = HipConv2d(2 * latent_channels, 2 * latent_channels, 1, padding=0)
        self.post_quant_conv = HipConv2d(latent_channels, latent_channels, 1, padding=0)
        # every 3x3 convolution of the encoder / decoder body feeds a GroupNorm (norm2, the next block's
        # norm1, the attention's group_norm or conv_norm_out): let it reduce that norm's statistics
        for name, m in self.named_modules():
            if isinstance(m, HipConv2d) and not name.endswith(("conv_shortcut", "conv_out", "quant_conv")):
                m.gn_groups = norm_num_groups

    @property
    def downsampling_factor(self) -> int:
        return 2 ** (len(self.config.block_out_channels) - 1)

    @classmethod
    def from_pretrained(cls, version: str, subfolder: str = "", revision: str = "main", allow_random_init: bool = False,
                        **kw) -> "AutoencoderKL":
        """Loads ``<version>/<subfolder>/config.json`` + ``*.safetensors`` from a LOCAL directory in the diffusers layout.
        There is no network here, so a hub name cannot be fetched: it raises ``FileNotFoundError`` — a model that silently
        carried untrained weights would produce garbage images with exit status 0.  ``allow_random_init=True`` (synthetic
        benchmarks, tests, or a caller that loads a full state dict right afterwards) builds the architecture of the
        named backbone (the SD 1.x and SDXL VAEs, the two the reference uses) with untrained weights instead."""
        path = os.path.join(version, subfolder) if subfolder else version
        if os.path.isdir(path):
            with open(os.path.join(path, "config.json")) as f:
                cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
            model = cls(**cfg)
            from safetensors.torch import load_file
            files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
            if not files:
                raise ValueError(f"no .safetensors weights under {path}")
            model.load_state_dict(load_file(os.path.join(path, files[0])), strict=True)
            return model
        if not allow_random_init:
            raise FileNotFoundError(
                f"VAE '{path}' is not a local diffusers-layout directory and hub downloads are unavailable. Pass a local "
                "directory, or allow_random_init=True to build the architecture with UNTRAINED weights (synthetic runs, "
                "or when a complete state dict is loaded afterwards).")
        return cls(**known_vae_config(version))

    # ---- API as used by the reference wrapper (autoencoderKL.py:64,116,134) -----------------
    @property
    def compute_dtype(self) -> torch.dtype:
        """activation dtype: f32 when the model is held in f32 (fp32 mode), bf16 otherwise"""
        return torch.float32 if self.quant_conv.weight.dtype == torch.float32 else torch.bfloat16

    def encode(self, x: torch.Tensor) -> AutoencoderKLOutput:
        xh = ops.nchw_to_nhwc(x, dtype=self.compute_dtype)
        h = self.encoder(xh)
        moments = self.quant_conv(h)
        return AutoencoderKLOutput(DiagonalGaussianDistribution(moments, self.config.latent_channels, x.dtype))

    def decode_nhwc(self, z_nhwc: torch.Tensor) -> torch.Tensor:
        """z_nhwc: bf16 [B, h, w, 8] (latent channels zero-padded to 8) -> bf16 NHWC image [B, 8h, 8w, 3]."""
        return self.decoder(self.post_quant_conv(z_nhwc))

    def decode(self, z: torch.Tensor) -> DecoderOutput:
        zh = ops.nchw_to_nhwc(z, dtype=self.compute_dtype)
        img = self.decode_nhwc(zh)
        return DecoderOutput(ops.nhwc_to_nchw(img, self.config.out_channels, out_dtype=z.dtype))
