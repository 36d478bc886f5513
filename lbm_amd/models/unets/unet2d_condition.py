"""UNet2DConditionModel on the MI355X kernels.

Module tree and parameter names are those of diffusers' ``UNet2DConditionModel`` as the
reference configures it (src/lbm/inference/utils.py:112-164 — SDXL-style, linear
projections, head count [5, 10, 20]; src/lbm/inference/relight.py:61-120 — SD1.5-style,
1x1-conv projections, 8 heads), so reference checkpoints load with ``strict=True``.
The forward pass is written for the hardware instead of being a module-by-module replay:

  * activations stay bf16 channels-last, so NHWC feature maps and token matrices are the
    same memory and every projection / convolution is one implicit-GEMM launch;
  * all per-resnet ``time_emb_proj`` layers are evaluated by ONE GEMM per forward;
  * bias, time-embedding add, residual add, SiLU and GEGLU run in GEMM epilogues;
  * nearest-x2 upsampling is folded into the following convolution's gather;
  * q and k come from one fused projection, V is produced already transposed for the
    flash-attention kernel.
"""
from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from ... import ops
from ..layers import HipConv2d, HipGroupNorm, HipLayerNorm, HipLinear, _Cached, fused_rows, training


class UNet2DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = HipLinear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = HipLinear(time_embed_dim, time_embed_dim)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, temb_channels: Optional[int], groups: int, eps: float):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = HipGroupNorm(groups, cin, eps=eps)
        self.conv1 = HipConv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = HipLinear(temb_channels, cout) if temb_channels else None
        self.norm2 = HipGroupNorm(groups, cout, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = HipConv2d(cout, cout, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = HipConv2d(cin, cout, 1, padding=0) if cin != cout else None

    def forward(self, x, temb_vec=None):
        """x: [B,H,W,Cin]; temb_vec: bf16 [B, Cout] = time_emb_proj(silu(temb)) (may be a strided view)."""
        h, x = self.norm1(x, silu=True, passthrough=True)      # x: the residual link's view of the input
        h = self.conv1(h, rowvec=temb_vec)
        h = self.norm2(h, silu=True)
        res = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return self.conv2(h, residual=res)


class Attention(nn.Module, _Cached):
    def __init__(self, query_dim: int, heads: int, dim_head: int, cross_attention_dim: Optional[int] = None,
                 bias: bool = False, out_bias: bool = True):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner_dim = heads, dim_head, inner
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = HipLinear(query_dim, inner, bias=bias)
        self.to_k = HipLinear(kv_dim, inner, bias=bias)
        self.to_v = HipLinear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([HipLinear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def _v_weight(self):
        def build():
            pw = ops.pack_linear_weight(self.to_v.weight, None)
            b = None if self.to_v.bias is None else self.to_v.bias.detach().float().contiguous()
            return pw.w, b
        return self._cached("v", (self.to_v.weight, self.to_v.bias), build)

    def _qkv_stack(self):
        """(stacked [3C, K] weight, column offset of q / k / v in the stack).  When the three weights lie back
        to back in one buffer (the trainer re-homes parameters into flat buffers) the stack is a view."""
        ws = sorted((("q", self.to_q.weight), ("k", self.to_k.weight), ("v", self.to_v.weight)),
                    key=lambda t: t[1].data_ptr())
        C, K = self.inner_dim, self.to_q.weight.shape[1]
        first = ws[0][1]
        adjacent = all(w.is_contiguous() and w.dtype == torch.bfloat16 and w.shape == first.shape and
                       w.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() for _, w in ws) and \
            all(ws[i][1].data_ptr() + C * K * 2 == ws[i + 1][1].data_ptr() for i in range(2))
        if adjacent:
            stack = torch.as_strided(first.detach(), (3 * C, K), (K, 1))
        else:
            stack = torch.cat([w.detach() for _, w in ws], 0)
        return stack, {r: i * C for i, (r, _) in enumerate(ws)}

    def packed_qkv(self):
        def build():
            stack, cols = self._qkv_stack()
            return ops.pack_linear_weight(stack, None), cols
        return self._cached("qkv", (self.to_q.weight, self.to_k.weight, self.to_v.weight), build)

    def packed_qkv_t(self):
        def build():
            stack, _ = self._qkv_stack()
            if stack.dtype == torch.bfloat16 and stack.shape[1] % 8 == 0:
                return ops.pack_linear_weight(ops.transpose(stack), None)
            return ops.pack_linear_weight(stack.t().contiguous(), None)
        return self._cached("qkvt", (self.to_q.weight, self.to_k.weight, self.to_v.weight), build)

    def forward(self, xn, ctx, residual):
        """xn: normalised tokens [B, N, C]; ctx None => self-attention over xn (the reference passes
        encoder_hidden_states=None with cross_attention_dim == C, so attn2 is a self-attention too);
        residual: tensor added to the output projection in the GEMM epilogue."""
        if self.dim_head not in (40, 64, 80, 160) and xn.dtype != torch.float32:
            raise NotImplementedError(f"flash attention kernels cover head_dim 40/64/80/160 (got {self.dim_head})")
        C = self.inner_dim
        if training(xn, self.to_q.weight):
            # differentiable route: separate projections, attention with a hand-written backward
            from ...autograd import AttentionFn, SelfAttnQKVFn
            if (ctx is None and self.to_q.bias is None and self.to_k.bias is None and self.to_v.bias is None
                    and self.to_q.weight.shape == self.to_k.weight.shape and xn.shape[-1] % 8 == 0
                    and ops.flash_attn_bwd_supported(self.dim_head, xn.shape[1], xn.shape[1])):
                o = SelfAttnQKVFn.apply(xn, self.to_q.weight, self.to_k.weight, self.to_v.weight, self,
                                        self.heads, self.dim_head, self.dim_head ** -0.5)
                return self.to_out[0](o, residual=residual)
            src = xn if ctx is None else ctx.to(torch.bfloat16).contiguous()
            q, k, v = self.to_q(xn), self.to_k(src), self.to_v(src)
            o = AttentionFn.apply(q, k, v, self.heads, self.dim_head, self.dim_head ** -0.5)
            return self.to_out[0](o, residual=residual)
        if ctx is None:
            qk = ops.linear(xn, fused_rows(self, "qk", [self.to_q, self.to_k], bias=self.to_q.bias is not None))
            wv, bv = self._v_weight()
            vt = ops.gemm_nt(wv, xn, bias_m=bv)            # [B, C, N]
            o = ops.flash_attn(qk[..., :C], qk[..., C:], vt, self.heads, self.dim_head)
        else:
            # a handful of context tokens (lighting parameters): K / V come from ctx, keys past the
            # token count are masked in the kernel; V^T is padded to a multiple of 8 keys with zeros
            ctx = ctx.to(xn.dtype).contiguous()
            M = ctx.shape[1]
            q = self.to_q(xn)
            k = self.to_k(ctx)
            v = self.to_v(ctx)                              # [B, M, C]
            vt = torch.zeros(ctx.shape[0], C, (M + 7) // 8 * 8, dtype=xn.dtype, device=ctx.device)
            vt[:, :, :M] = v.transpose(1, 2)
            o = ops.flash_attn(q, k, vt, self.heads, self.dim_head)
        return self.to_out[0](o, residual=residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = HipLinear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), HipLinear(dim * mult, dim)])

    def forward(self, xn, residual):
        if training(xn, self.net[2].weight):
            from ...autograd import GegluFn
            return self.net[2](GegluFn.apply(self.net[0].proj(xn)), residual=residual)
        return self.net[2](self.net[0].proj(xn, geglu=True), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.norm1 = HipLayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head, None)
        self.norm2 = HipLayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim)
        self.norm3 = HipLayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx):
        xn, h = self.norm1(h, passthrough=True)      # h: the residual link's view of the block input (layers.HipLayerNorm)
        h = self.attn1(xn, None, h)
        xn, h = self.norm2(h, passthrough=True)
        h = self.attn2(xn, ctx, h)
        xn, h = self.norm3(h, passthrough=True)
        return self.ff(xn, h)


class Transformer2DModel(nn.Module):
    def __init__(self, heads: int, dim_head: int, in_channels: int, num_layers: int, cross_attention_dim: int,
                 groups: int, use_linear_projection: bool):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = HipGroupNorm(groups, in_channels, eps=1e-6)
        if use_linear_projection:
            self.proj_in = HipLinear(in_channels, inner)
            self.proj_out = HipLinear(inner, in_channels)
        else:
            self.proj_in = HipConv2d(in_channels, inner, 1, padding=0)
            self.proj_out = HipConv2d(inner, in_channels, 1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])

    def forward(self, x, ctx):
        B, H, W, C = x.shape
        h, x = self.norm(x, passthrough=True)
        # a 1x1 convolution on NHWC and a linear layer on tokens are the same GEMM
        h = self.proj_in(h)
        h = h.view(B, H * W, -1)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.use_linear_projection:
            out = self.proj_out(h, residual=x.view(B, H * W, C))
        else:
            out = self.proj_out(h.view(B, H, W, -1), residual=x)
        return out.view(B, H, W, C)


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        self.conv = HipConv2d(channels, channels, 3, stride=2, padding=padding, asym_pad=(padding == 0))

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = HipConv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(x, up2=True)       # nearest x2 folded into the conv's gather


class _Block(nn.Module):
    """One resolution level: resnets, optional transformers, optional resampler."""

    def __init__(self, resnets, attentions, downsamplers=None, upsamplers=None):
        super().__init__()
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        if downsamplers is not None:
            self.downsamplers = nn.ModuleList(downsamplers)
        if upsamplers is not None:
            self.upsamplers = nn.ModuleList(upsamplers)

    @property
    def has_attn(self):
        return hasattr(self, "attentions")


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


class UNet2DConditionModel(nn.Module, _Cached):
    def __init__(self, in_channels=4, out_channels=4,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type="UNetMidBlock2DCrossAttn",
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280, transformer_layers_per_block=1,
                 attention_head_dim=8, num_attention_heads=None, use_linear_projection=False,
                 flip_sin_to_cos=True, freq_shift=0, center_input_sample=False, act_fn="silu", **unused):
        super().__init__()
        chans = list(block_out_channels)
        n = len(chans)
        if act_fn != "silu" or not flip_sin_to_cos or freq_shift != 0 or center_input_sample:
            raise NotImplementedError("only act_fn='silu', flip_sin_to_cos=True, freq_shift=0, center_input_sample=False")
        if mid_block_type != "UNetMidBlock2DCrossAttn":
            raise NotImplementedError(mid_block_type)
        # diffusers: num_attention_heads defaults to attention_head_dim, i.e. the list is the HEAD COUNT
        heads = _as_list(num_attention_heads if num_attention_heads is not None else attention_head_dim, n)
        tlpb = _as_list(transformer_layers_per_block, n)
        cad = _as_list(cross_attention_dim, n)
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, down_block_types=list(down_block_types),
            mid_block_type=mid_block_type, up_block_types=list(up_block_types), block_out_channels=chans,
            layers_per_block=layers_per_block, downsample_padding=downsample_padding,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
            transformer_layers_per_block=transformer_layers_per_block, attention_head_dim=attention_head_dim,
            num_attention_heads=num_attention_heads, use_linear_projection=use_linear_projection,
            flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift, center_input_sample=center_input_sample,
            act_fn=act_fn, **unused)
        G, eps, lin = norm_num_groups, norm_eps, use_linear_projection
        temb_ch = chans[0] * 4
        self.temb_in = chans[0]

        self.conv_in = HipConv2d(in_channels, chans[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(chans[0], temb_ch)

        def tf(c, i_heads, depth, ca):
            return Transformer2DModel(i_heads, c // i_heads, c, depth, ca, G, lin)

        self.down_blocks = nn.ModuleList()
        out_c = chans[0]
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, chans[i]
            last = i == n - 1
            resnets = [ResnetBlock2D(in_c if j == 0 else out_c, out_c, temb_ch, G, eps) for j in range(layers_per_block)]
            attns = None
            if kind == "CrossAttnDownBlock2D":
                attns = [tf(out_c, heads[i], tlpb[i], cad[i]) for _ in range(layers_per_block)]
            elif kind != "DownBlock2D":
                raise NotImplementedError(kind)
            self.down_blocks.append(_Block(resnets, attns,
                                           downsamplers=None if last else [Downsample2D(out_c, downsample_padding)]))

        mid_c = chans[-1]
        self.mid_block = _Block([ResnetBlock2D(mid_c, mid_c, temb_ch, G, eps) for _ in range(2)],
                                [tf(mid_c, heads[-1], tlpb[-1], cad[-1])])

        self.up_blocks = nn.ModuleList()
        rch, rheads, rtl, rcad = chans[::-1], heads[::-1], tlpb[::-1], cad[::-1]
        out_c = rch[0]
        for i, kind in enumerate(up_block_types):
            prev_c, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, n - 1)]
            last = i == n - 1
            resnets = []
            for j in range(layers_per_block + 1):
                skip_c = in_c if j == layers_per_block else out_c
                rin = prev_c if j == 0 else out_c
                resnets.append(ResnetBlock2D(rin + skip_c, out_c, temb_ch, G, eps))
            attns = None
            if kind == "CrossAttnUpBlock2D":
                attns = [tf(out_c, rheads[i], rtl[i], rcad[i]) for _ in range(layers_per_block + 1)]
            elif kind != "UpBlock2D":
                raise NotImplementedError(kind)
            self.up_blocks.append(_Block(resnets, attns, upsamplers=None if last else [Upsample2D(out_c)]))

        self.conv_norm_out = HipGroupNorm(G, chans[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = HipConv2d(chans[0], out_channels, 3, padding=1)

    @property
    def compute_dtype(self) -> torch.dtype:
        """activation dtype: f32 when the model is held in f32 (fp32 mode), bf16 otherwise"""
        return torch.float32 if self.conv_in.weight.dtype == torch.float32 else torch.bfloat16

    # ---- time embedding: every resnet's projection in one GEMM -----------------------------
    def _resnets(self) -> List[ResnetBlock2D]:
        rs = []
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            rs += list(blk.resnets)
        return rs

    def _time_vectors(self, timestep: torch.Tensor, B: int, device):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=device)
        t = t.to(device=device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t_emb = ops.timestep_embedding(t.contiguous(), self.temb_in, dtype=self.compute_dtype)
        # differentiable route when ANY parameter on the time path trains (the embedding MLP or a resnet's projection):
        # a single sentinel would silently starve the others of their gradient
        time_params = list(self.time_embedding.parameters()) + [p for r in self._resnets() for p in r.time_emb_proj.parameters()]
        if training(*time_params):
            from ...autograd import SiluFn
            e = SiluFn.apply(self.time_embedding.linear_2(SiluFn.apply(self.time_embedding.linear_1(t_emb))))
            return {id(r): r.time_emb_proj(e) for r in self._resnets()}
        e = self.time_embedding.linear_1(t_emb, act=ops.ACT_SILU)
        e = self.time_embedding.linear_2(e, act=ops.ACT_SILU)        # silu(temb): the only form the resnets consume
        resnets = self._resnets()
        allp = ops.linear(e, fused_rows(self, "temb_all", [r.time_emb_proj for r in resnets], bias=True))
        vecs, off = {}, 0
        for r in resnets:
            vecs[id(r)] = allp[:, off:off + r.out_channels]
            off += r.out_channels
        return vecs

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: Optional[torch.Tensor] = None,
                class_labels=None, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                return_dict: bool = True, **kwargs):
        """sample: NCHW (bf16 or f32); returns an object with ``.sample`` NCHW in sample.dtype."""
        if class_labels is not None or added_cond_kwargs is not None or down_block_additional_residuals is not None \
                or mid_block_additional_residual is not None or down_intrablock_additional_residuals is not None:
            raise NotImplementedError("class labels / added conditions / ControlNet residuals are not used by the reference's builders")
        B = sample.shape[0]
        # differentiable route when the input or ANY parameter of the model wants a gradient (a trainable_params
        # pattern may freeze conv_in while the blocks train)
        train = torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in self.parameters()))
        tv = self._time_vectors(timestep, B, sample.device)
        if train and sample.requires_grad:
            from ...autograd import NchwToNhwcFn
            x = NchwToNhwcFn.apply(sample)
        else:
            x = ops.nchw_to_nhwc(sample, dtype=self.compute_dtype)
        ctx = encoder_hidden_states
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(h, tv[id(res)])
                if blk.has_attn:
                    h = blk.attentions[j](h, ctx)
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0](h)
                skips.append(h)
        m = self.mid_block
        h = m.resnets[0](h, tv[id(m.resnets[0])])
        h = m.attentions[0](h, ctx)
        h = m.resnets[1](h, tv[id(m.resnets[1])])
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                h = torch.cat([h, skips.pop()], dim=-1) if train else ops.concat_channels(h, skips.pop())
                h = res(h, tv[id(res)])
                if blk.has_attn:
                    h = blk.attentions[j](h, ctx)
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0](h)
        h = self.conv_norm_out(h, silu=True)
        h = self.conv_out(h)
        if train:
            from ...autograd import NhwcToNchwFn
            out = NhwcToNchwFn.apply(h, self.config.out_channels, sample.dtype)
        else:
            out = ops.nhwc_to_nchw(h, self.config.out_channels, out_dtype=sample.dtype)
        return UNet2DConditionOutput(out) if return_dict else (out,)
