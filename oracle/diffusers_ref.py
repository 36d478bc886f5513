"""TEST INFRASTRUCTURE — CPU fp32 oracle, never imported by the product path.

Plain-PyTorch (torch.nn.functional, fp32) restatement of the three diffusers 0.32.2
components the reference delegates its arithmetic to (requirements.txt:2 of the
reference pins ``diffusers==0.32.2``; the package is NOT vendored in the reference tree
and NOT installed here):

  * ``UNet2DConditionModel``        — called at src/lbm/models/unets/unet.py:127-140,
                                      configured at src/lbm/inference/utils.py:112-164 and
                                      src/lbm/inference/relight.py:61-120
  * ``AutoencoderKL``               — called at src/lbm/models/vae/autoencoderKL.py:64,116,134
  * ``FlowMatchEulerDiscreteScheduler`` — src/lbm/models/lbm/lbm_model.py:320,346,
                                      built at src/lbm/inference/utils.py:209-212

PARITY UNPINNED at this boundary: the reference holds no tests, golden vectors or
checkpoints for these modules, and diffusers cannot be imported or fetched here, so this
file restates diffusers' published forward passes from the module definitions of 0.32.2.
It is pinned only structurally (state-dict key names / shapes, exact parameter counts
859,520,964 for the canonical SD1.5 UNet and 83,653,863 for the SD VAE — see
tests/test_structure.py) and by closed-form checks of the scheduler.

Every function is functional: it takes a flat ``state_dict`` with diffusers key names
(so the product's ``state_dict()`` feeds it directly) and fp32 NCHW tensors.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# Arithmetic dtype of the restatement.  fp32 is the oracle of record; bfloat16 reproduces how the
# reference actually executes (every builder casts the model with ``.to(torch.bfloat16)``, e.g.
# src/lbm/inference/utils.py:164,220), which tests use to put the bf16 error of the HIP path in context.
_COMPUTE = [torch.float32]


class compute_dtype:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        _COMPUTE.append(self.dtype)

    def __exit__(self, *a):
        _COMPUTE.pop()


def _p(sd: SD, key: str) -> torch.Tensor:
    return sd[key].to(_COMPUTE[-1])


def _has(sd: SD, key: str) -> bool:
    return key in sd


# ---------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``:
    cos half first (src/lbm/inference/utils.py:116-117 sets flip_sin_to_cos / freq_shift)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps.float()[:, None] * torch.exp(exponent).to(timesteps.device)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1).to(_COMPUTE[-1])


def conv(sd: SD, prefix: str, x, stride=1, padding=1):
    b = _p(sd, prefix + ".bias") if _has(sd, prefix + ".bias") else None
    return F.conv2d(x, _p(sd, prefix + ".weight"), b, stride=stride, padding=padding)


def linear(sd: SD, prefix: str, x):
    b = _p(sd, prefix + ".bias") if _has(sd, prefix + ".bias") else None
    return F.linear(x, _p(sd, prefix + ".weight"), b)


def group_norm(sd: SD, prefix: str, x, groups: int, eps: float):
    return F.group_norm(x, groups, _p(sd, prefix + ".weight"), _p(sd, prefix + ".bias"), eps)


def layer_norm(sd: SD, prefix: str, x, eps: float = 1e-5):
    w = _p(sd, prefix + ".weight")
    return F.layer_norm(x, (w.shape[0],), w, _p(sd, prefix + ".bias"), eps)


def resnet_block(sd: SD, prefix: str, x, temb: Optional[torch.Tensor], groups: int, eps: float):
    """diffusers ResnetBlock2D, time_embedding_norm='default', output_scale_factor=1."""
    h = F.silu(group_norm(sd, prefix + ".norm1", x, groups, eps))
    h = conv(sd, prefix + ".conv1", h)
    if temb is not None and _has(sd, prefix + ".time_emb_proj.weight"):
        h = h + linear(sd, prefix + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(sd, prefix + ".norm2", h, groups, eps))
    h = conv(sd, prefix + ".conv2", h)
    if _has(sd, prefix + ".conv_shortcut.weight"):
        x = conv(sd, prefix + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd: SD, prefix: str, x, ctx: Optional[torch.Tensor], heads: int):
    """diffusers Attention + AttnProcessor2_0 on (B, N, C) tokens; ctx None => self-attention."""
    if ctx is None:
        ctx = x
    q = linear(sd, prefix + ".to_q", x)
    k = linear(sd, prefix + ".to_k", ctx)
    v = linear(sd, prefix + ".to_v", ctx)
    B, N, C = q.shape
    d = C // heads
    q = q.view(B, N, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return linear(sd, prefix + ".to_out.0", o)


def feed_forward(sd: SD, prefix: str, x):
    """FeedForward(activation_fn='geglu'): value half first, gate half second, erf GELU."""
    h = linear(sd, prefix + ".net.0.proj", x)
    val, gate = h.chunk(2, dim=-1)
    return linear(sd, prefix + ".net.2", val * F.gelu(gate))


def basic_transformer_block(sd: SD, prefix: str, x, ctx, heads: int):
    x = x + attention(sd, prefix + ".attn1", layer_norm(sd, prefix + ".norm1", x), None, heads)
    x = x + attention(sd, prefix + ".attn2", layer_norm(sd, prefix + ".norm2", x), ctx, heads)
    x = x + feed_forward(sd, prefix + ".ff", layer_norm(sd, prefix + ".norm3", x))
    return x


def transformer_2d(sd: SD, prefix: str, x, ctx, heads: int, groups: int):
    """diffusers Transformer2DModel (continuous input). The projection kind (Linear vs 1x1
    conv, ``use_linear_projection``) is read off the weight rank."""
    B, C, H, W = x.shape
    res = x
    h = group_norm(sd, prefix + ".norm", x, groups, 1e-6)
    lin = _p(sd, prefix + ".proj_in.weight").dim() == 2
    if lin:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = linear(sd, prefix + ".proj_in", h)
    else:
        h = conv(sd, prefix + ".proj_in", h, padding=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    k = 0
    while _has(sd, f"{prefix}.transformer_blocks.{k}.norm1.weight"):
        h = basic_transformer_block(sd, f"{prefix}.transformer_blocks.{k}", h, ctx, heads)
        k += 1
    if lin:
        h = linear(sd, prefix + ".proj_out", h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = conv(sd, prefix + ".proj_out", h, padding=0)
    return h + res


def _count(sd: SD, pattern: str) -> int:
    n = 0
    while _has(sd, pattern.format(n)):
        n += 1
    return n


# ---------------------------------------------------------------------------------------
# UNet2DConditionModel
# ---------------------------------------------------------------------------------------
def unet_forward(sd: SD, cfg: dict, sample, timestep, encoder_hidden_states=None):
    """cfg keys (diffusers names): block_out_channels, attention_head_dim (used as the HEAD
    COUNT per level because num_attention_heads is None — src/lbm/inference/utils.py:139-140),
    norm_num_groups, norm_eps.  Block structure is read from the state-dict keys."""
    groups, eps = cfg.get("norm_num_groups", 32), cfg.get("norm_eps", 1e-5)
    chans = list(cfg["block_out_channels"])
    heads = cfg["attention_head_dim"]
    heads = list(heads) if isinstance(heads, (list, tuple)) else [heads] * len(chans)
    B = sample.shape[0]
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float32)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(B)
    temb = timestep_embedding(t, chans[0])
    temb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", temb)))

    h = conv(sd, "conv_in", sample)
    skips = [h]
    n_down = len(chans)
    for i in range(n_down):
        p = f"down_blocks.{i}"
        nres = _count(sd, p + ".resnets.{}.norm1.weight")
        has_attn = _has(sd, p + ".attentions.0.norm.weight")
        for j in range(nres):
            h = resnet_block(sd, f"{p}.resnets.{j}", h, temb, groups, eps)
            if has_attn:
                h = transformer_2d(sd, f"{p}.attentions.{j}", h, encoder_hidden_states, heads[i], groups)
            skips.append(h)
        if _has(sd, p + ".downsamplers.0.conv.weight"):
            h = conv(sd, p + ".downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)

    h = resnet_block(sd, "mid_block.resnets.0", h, temb, groups, eps)
    h = transformer_2d(sd, "mid_block.attentions.0", h, encoder_hidden_states, heads[-1], groups)
    h = resnet_block(sd, "mid_block.resnets.1", h, temb, groups, eps)

    rheads = heads[::-1]
    for i in range(n_down):
        p = f"up_blocks.{i}"
        nres = _count(sd, p + ".resnets.{}.norm1.weight")
        has_attn = _has(sd, p + ".attentions.0.norm.weight")
        for j in range(nres):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"{p}.resnets.{j}", h, temb, groups, eps)
            if has_attn:
                h = transformer_2d(sd, f"{p}.attentions.{j}", h, encoder_hidden_states, rheads[i], groups)
        if _has(sd, p + ".upsamplers.0.conv.weight"):
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, p + ".upsamplers.0.conv", h)
    assert not skips
    h = F.silu(group_norm(sd, "conv_norm_out", h, groups, eps))
    return conv(sd, "conv_out", h)


# ---------------------------------------------------------------------------------------
# UNet2DModel (pixel-space refine model, src/lbm/inference/relight.py:356-366)
# ---------------------------------------------------------------------------------------
def _spatial_attention(sd: SD, prefix: str, x, heads: int, groups: int, eps: float):
    """diffusers Attention(residual_connection=True, bias=True, norm_num_groups=groups) on a feature map
    (the deprecated-attention-block form used by UNetMidBlock2D and the VAE mid block)."""
    B, C, H, W = x.shape
    h = group_norm(sd, prefix + ".group_norm", x, groups, eps)
    tok = h.view(B, C, H * W).transpose(1, 2)
    o = attention(sd, prefix, tok, None, heads)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def unet2d_forward(sd: SD, cfg: dict, sample, timestep):
    """diffusers 0.32.2 UNet2DModel.forward with DownBlock2D / UpBlock2D levels, positional time embedding,
    UNetMidBlock2D with one self-attention of C / attention_head_dim heads (default head dim 8)."""
    chans = list(cfg["block_out_channels"])
    groups, eps = cfg.get("norm_num_groups", 32), cfg.get("norm_eps", 1e-5)
    lpb = cfg.get("layers_per_block", 2)
    head_dim = cfg.get("attention_head_dim", 8)
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).float().expand(sample.shape[0]) if t.numel() == 1 else t.reshape(-1).float()
    temb = linear(sd, "time_embedding.linear_1", timestep_embedding(t, chans[0]))
    temb = linear(sd, "time_embedding.linear_2", F.silu(temb))
    h = conv(sd, "conv_in", sample.to(_COMPUTE[-1]))
    skips = [h]
    for i in range(len(chans)):
        for j in range(lpb):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, temb, groups, eps)
            skips.append(h)
        if _has(sd, f"down_blocks.{i}.downsamplers.0.conv.weight"):
            h = conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    h = resnet_block(sd, "mid_block.resnets.0", h, temb, groups, eps)
    if _has(sd, "mid_block.attentions.0.to_q.weight"):
        h = _spatial_attention(sd, "mid_block.attentions.0", h, chans[-1] // head_dim, groups, eps)
    h = resnet_block(sd, "mid_block.resnets.1", h, temb, groups, eps)
    for i in range(len(chans)):
        for j in range(lpb + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, temb, groups, eps)
        if _has(sd, f"up_blocks.{i}.upsamplers.0.conv.weight"):
            h = conv(sd, f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    h = F.silu(group_norm(sd, "conv_norm_out", h, groups, eps))
    return conv(sd, "conv_out", h)


# ---------------------------------------------------------------------------------------
# AutoencoderKL
# ---------------------------------------------------------------------------------------
VAE_EPS = 1e-6
VAE_GROUPS = 32


def _vae_mid_attention(sd: SD, prefix: str, x):
    """Attention(heads=1, dim_head=C, residual_connection=True, bias=True, norm_num_groups=32)."""
    B, C, H, W = x.shape
    h = group_norm(sd, prefix + ".group_norm", x, VAE_GROUPS, VAE_EPS)
    tok = h.view(B, C, H * W).transpose(1, 2)
    q = linear(sd, prefix + ".to_q", tok)[:, None]
    k = linear(sd, prefix + ".to_k", tok)[:, None]
    v = linear(sd, prefix + ".to_v", tok)[:, None]
    o = F.scaled_dot_product_attention(q, k, v)[:, 0]
    o = linear(sd, prefix + ".to_out.0", o)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def _vae_mid(sd: SD, prefix: str, h):
    h = resnet_block(sd, prefix + ".resnets.0", h, None, VAE_GROUPS, VAE_EPS)
    h = _vae_mid_attention(sd, prefix + ".attentions.0", h)
    return resnet_block(sd, prefix + ".resnets.1", h, None, VAE_GROUPS, VAE_EPS)


def vae_encode_moments(sd: SD, x):
    """AutoencoderKL.encode up to the posterior parameters: (B, 2*latent, H/8, W/8)."""
    h = conv(sd, "encoder.conv_in", x)
    i = 0
    while _has(sd, f"encoder.down_blocks.{i}.resnets.0.norm1.weight"):
        p = f"encoder.down_blocks.{i}"
        for j in range(_count(sd, p + ".resnets.{}.norm1.weight")):
            h = resnet_block(sd, f"{p}.resnets.{j}", h, None, VAE_GROUPS, VAE_EPS)
        if _has(sd, p + ".downsamplers.0.conv.weight"):
            h = F.pad(h, (0, 1, 0, 1))                      # asymmetric: right / bottom only
            h = conv(sd, p + ".downsamplers.0.conv", h, stride=2, padding=0)
        i += 1
    h = _vae_mid(sd, "encoder.mid_block", h)
    h = F.silu(group_norm(sd, "encoder.conv_norm_out", h, VAE_GROUPS, VAE_EPS))
    h = conv(sd, "encoder.conv_out", h)
    return conv(sd, "quant_conv", h, padding=0)


def posterior_sample(moments, eps):
    """DiagonalGaussianDistribution.sample(): mean + exp(0.5*clamp(logvar,-30,20)) * eps."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * eps


def vae_decode(sd: SD, z):
    h = conv(sd, "post_quant_conv", z, padding=0)
    h = conv(sd, "decoder.conv_in", h)
    h = _vae_mid(sd, "decoder.mid_block", h)
    i = 0
    while _has(sd, f"decoder.up_blocks.{i}.resnets.0.norm1.weight"):
        p = f"decoder.up_blocks.{i}"
        for j in range(_count(sd, p + ".resnets.{}.norm1.weight")):
            h = resnet_block(sd, f"{p}.resnets.{j}", h, None, VAE_GROUPS, VAE_EPS)
        if _has(sd, p + ".upsamplers.0.conv.weight"):
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, p + ".upsamplers.0.conv", h)
        i += 1
    h = F.silu(group_norm(sd, "decoder.conv_norm_out", h, VAE_GROUPS, VAE_EPS))
    return conv(sd, "decoder.conv_out", h)


# ---------------------------------------------------------------------------------------
# FlowMatchEulerDiscreteScheduler (shift = 1, no dynamic shifting)
# ---------------------------------------------------------------------------------------
class FlowMatchEulerOracle:
    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self._step_index = None

    def set_timesteps(self, sigmas):
        s = np.array(sigmas).astype(np.float32)
        s = self.shift * s / (1 + (self.shift - 1) * s)
        s = torch.from_numpy(s)
        self.timesteps = s * self.num_train_timesteps
        self.sigmas = torch.cat([s, torch.zeros(1)])
        self._step_index = None

    def step(self, model_output, timestep, sample):
        if self._step_index is None:
            idx = (self.timesteps == timestep).nonzero()
            self._step_index = idx[1 if len(idx) > 1 else 0].item()
        s, s_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = sample.to(torch.float32) + (s_next - s) * model_output
        self._step_index += 1
        return prev.to(model_output.dtype)
