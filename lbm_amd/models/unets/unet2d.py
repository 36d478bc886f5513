"""UNet2DModel with diffusers' module tree and state-dict names, forward on the HIP kernels: the
pixel-space refine denoiser of the reference (DownBlock2D / UpBlock2D levels, positional time embedding,
UNetMidBlock2D with one self-attention of C / attention_head_dim heads; src/lbm/inference/relight.py:356-366,
src/lbm/models/unets/unet.py:7-52).  It shares every building block with the conditional UNet."""
from types import SimpleNamespace
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from ... import ops
from ..layers import HipConv2d, HipGroupNorm, _Cached, fused_rows, training
from .unet2d_condition import (Downsample2D, ResnetBlock2D, TimestepEmbedding, UNet2DConditionOutput, Upsample2D,
                               _Block)


class UNet2DModel(nn.Module, _Cached):
    def __init__(self, sample_size=None, in_channels=3, out_channels=3, center_input_sample=False,
                 time_embedding_type="positional", freq_shift=0, flip_sin_to_cos=True,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1,
                 downsample_padding=1, act_fn="silu", attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 add_attention=True, **unused):
        super().__init__()
        if any(k != "DownBlock2D" for k in down_block_types) or any(k != "UpBlock2D" for k in up_block_types):
            raise NotImplementedError("only DownBlock2D / UpBlock2D levels (the reference's refine model)")
        if act_fn != "silu" or time_embedding_type != "positional" or not flip_sin_to_cos or freq_shift != 0 \
                or center_input_sample or mid_block_scale_factor != 1:
            raise NotImplementedError("only act_fn='silu', positional time embedding (cos first, no shift), scale factor 1")
        chans = list(block_out_channels)
        n = len(chans)
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels, center_input_sample=False,
            time_embedding_type=time_embedding_type, freq_shift=freq_shift, flip_sin_to_cos=flip_sin_to_cos,
            down_block_types=list(down_block_types), up_block_types=list(up_block_types), block_out_channels=chans,
            layers_per_block=layers_per_block, mid_block_scale_factor=mid_block_scale_factor,
            downsample_padding=downsample_padding, act_fn=act_fn, attention_head_dim=attention_head_dim,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, add_attention=add_attention, **unused)
        G, eps = norm_num_groups, norm_eps
        temb_ch = chans[0] * 4
        self.temb_in = chans[0]
        self.conv_in = HipConv2d(in_channels, chans[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(chans[0], temb_ch)
        self.down_blocks = nn.ModuleList()
        out_c = chans[0]
        for i in range(n):
            in_c, out_c = out_c, chans[i]
            resnets = [ResnetBlock2D(in_c if j == 0 else out_c, out_c, temb_ch, G, eps) for j in range(layers_per_block)]
            self.down_blocks.append(_Block(resnets, None, downsamplers=None if i == n - 1 else [Downsample2D(out_c, downsample_padding)]))
        mid_c = chans[-1]
        from ..vae.autoencoder_kl import VaeAttention      # deferred: the VAE module imports this package's blocks
        attn = [VaeAttention(mid_c, G, heads=mid_c // attention_head_dim, eps=eps)] if add_attention else None
        self.mid_block = _Block([ResnetBlock2D(mid_c, mid_c, temb_ch, G, eps) for _ in range(2)], attn)
        self.up_blocks = nn.ModuleList()
        rch = chans[::-1]
        out_c = rch[0]
        for i in range(n):
            prev_c, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, n - 1)]
            resnets = []
            for j in range(layers_per_block + 1):
                skip_c = in_c if j == layers_per_block else out_c
                resnets.append(ResnetBlock2D((prev_c if j == 0 else out_c) + skip_c, out_c, temb_ch, G, eps))
            self.up_blocks.append(_Block(resnets, None, upsamplers=None if i == n - 1 else [Upsample2D(out_c)]))
        self.conv_norm_out = HipGroupNorm(G if G is not None else min(chans[0] // 4, 32), chans[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = HipConv2d(chans[0], out_channels, 3, padding=1)

    def _resnets(self) -> List[ResnetBlock2D]:
        rs = []
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            rs += list(blk.resnets)
        return rs

    def _time_vectors(self, timestep, B: int, device):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=device)
        t = t.to(device=device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t_emb = ops.timestep_embedding(t.contiguous(), self.temb_in, dtype=self.conv_in.weight.dtype if self.conv_in.weight.dtype == torch.float32 else torch.bfloat16)
        time_params = list(self.time_embedding.parameters()) + [p for r in self._resnets() for p in r.time_emb_proj.parameters()]
        if training(*time_params):      # any trainable parameter on the time path
            from ...autograd import SiluFn
            e = SiluFn.apply(self.time_embedding.linear_2(SiluFn.apply(self.time_embedding.linear_1(t_emb))))
            return {id(r): r.time_emb_proj(e) for r in self._resnets()}
        e = self.time_embedding.linear_1(t_emb, act=ops.ACT_SILU)
        e = self.time_embedding.linear_2(e, act=ops.ACT_SILU)
        resnets = self._resnets()
        allp = ops.linear(e, fused_rows(self, "temb_all", [r.time_emb_proj for r in resnets], bias=True))
        vecs, off = {}, 0
        for r in resnets:
            vecs[id(r)] = allp[:, off:off + r.out_channels]
            off += r.out_channels
        return vecs

    def forward(self, sample: torch.Tensor, timestep, class_labels=None, return_dict: bool = True):
        """sample: NCHW (bf16 or f32); returns an object with ``.sample`` NCHW in sample.dtype."""
        if class_labels is not None:
            raise NotImplementedError("class conditioning is not used by the reference's builders")
        B = sample.shape[0]
        train = torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in self.parameters()))
        tv = self._time_vectors(timestep, B, sample.device)
        if train and sample.requires_grad:
            from ...autograd import NchwToNhwcFn
            x = NchwToNhwcFn.apply(sample)
        else:
            x = ops.nchw_to_nhwc(sample, dtype=torch.float32 if self.conv_in.weight.dtype == torch.float32 else torch.bfloat16)
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            for res in blk.resnets:
                h = res(h, tv[id(res)])
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0](h)
                skips.append(h)
        m = self.mid_block
        h = m.resnets[0](h, tv[id(m.resnets[0])])
        if m.has_attn:
            h = m.attentions[0](h)
        h = m.resnets[1](h, tv[id(m.resnets[1])])
        for blk in self.up_blocks:
            for res in blk.resnets:
                h = torch.cat([h, skips.pop()], dim=-1) if train else ops.concat_channels(h, skips.pop())
                h = res(h, tv[id(res)])
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0](h)
        h = self.conv_out(self.conv_norm_out(h, silu=True))
        if train:
            from ...autograd import NhwcToNchwFn
            out = NhwcToNchwFn.apply(h, self.config.out_channels, sample.dtype)
        else:
            out = ops.nhwc_to_nchw(h, self.config.out_channels, out_dtype=sample.dtype)
        return UNet2DConditionOutput(out) if return_dict else (out,)


class DiffusersUNet2DWrapper(UNet2DModel):
    """Same call protocol as the reference's wrapper (src/lbm/models/unets/unet.py:7-52)."""

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                conditioning: Dict[str, torch.Tensor] = None, *args, **kwargs):
        class_labels = concat = None
        if conditioning is not None:
            class_labels = conditioning["cond"].get("vector", None)
            concat = conditioning["cond"].get("concat", None)
        if concat is not None:
            sample = torch.cat([sample, concat.to(sample.dtype)], dim=1)
        return UNet2DModel.forward(self, sample, timestep, class_labels).sample

    def freeze(self):
        self.eval()
        for param in self.parameters():
            param.requires_grad = False
