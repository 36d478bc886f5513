"""Parameter-holding layers whose forward runs on the HIP kernels.

Each class subclasses the torch.nn layer diffusers uses, so parameter names, shapes and
default initialisation are identical (checkpoints load with ``strict=True``); ``forward``
never calls torch's implementation.  Activations are bf16 channels-last ``[B, H, W, C]``
(equivalently ``[B, H*W, C]`` tokens).  Packed / converted weights are cached per layer
and rebuilt when the parameter is modified in place, replaced, or moved.
"""
import os
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


# LBM_NO_UP2_SPLIT=1: upsampling convolutions gather through the upsample (3x3, 9 taps) instead of running as four 2x2
# parity convolutions (same-box A/B runs)
_UP2_SPLIT = os.environ.get("LBM_NO_UP2_SPLIT") is None


def _stamp(*tensors):
    return tuple((t.data_ptr(), t._version, t.device, t.dtype) if t is not None else None for t in tensors)


class _Cached:
    """Small per-module cache keyed on the identity/version of the source parameters."""

    def _cached(self, name, params, build):
        """The cached value, built on a miss.  A build enqueues kernels on the CURRENT stream while the entry
        becomes visible to the host at once, so another stream (the two batch halves of lbm_amd.streams run the
        same layers on two streams) could launch a consumer before the pack is written.  An entry built on the GPU
        therefore carries an event recorded behind its build: a hit from any other stream waits on it, until the
        event has completed (after that the pack is plain resident memory for every stream)."""
        key = _stamp(*params)
        ent = self.__dict__.get("_pack_" + name)
        if ent is None or ent[0] != key:
            with torch.no_grad():
                val = build()
            ent = [key, val, None, 0, None]              # key, value, build event, build stream, device
            p0 = next((p for p in params if p is not None), None)
            if p0 is not None and p0.is_cuda and not torch.cuda.is_current_stream_capturing():
                s = torch.cuda.current_stream(p0.device)
                ent[2], ent[3], ent[4] = s.record_event(), s.cuda_stream, p0.device
            self.__dict__["_pack_" + name] = ent
        elif ent[2] is not None and not torch.cuda.is_current_stream_capturing():
            if ent[2].query():
                ent[2] = None
            else:
                s = torch.cuda.current_stream(ent[4])
                if s.cuda_stream != ent[3]:
                    s.wait_event(ent[2])
        return ent[1]


class HipConv2d(nn.Conv2d, _Cached):
    """3x3 / 1x1 convolution as implicit GEMM. ``asym_pad``: pad right/bottom only (VAE encoder
    downsample: F.pad(x, (0, 1, 0, 1)) + stride-2 conv with padding 0)."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1, asym_pad=False):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=0 if asym_pad else padding)
        self.asym_pad = asym_pad
        # groups of the GroupNorm that consumes this layer's output (0: none): the convolution then
        # reduces the statistics in its epilogue and the norm skips its own pass over the tensor
        self.gn_groups = 0

    def packed(self) -> ops.PackedWeight:
        return self._cached("w", (self.weight, self.bias), lambda: ops.pack_conv_weight(self.weight, self.bias))

    def packed_up2(self):
        """the four parity filters of 'nearest x2 upsample, then this 3x3 convolution' (ops.pack_conv_weight_up2)"""
        return self._cached("u", (self.weight, self.bias), lambda: ops.pack_conv_weight_up2(self.weight, self.bias))

    def packed_dgrad(self) -> ops.PackedWeight:
        """input-gradient filter of a stride-1 convolution: in/out channels swapped, taps flipped"""
        return self._cached("d", (self.weight,), lambda: ops.pack_conv_weight_dgrad(self.weight))

    def packed_col_t(self) -> ops.PackedWeight:
        """[KH*KW*Cin_p, Cout_p] matrix (tap-major rows): dcol = dY . W for the col2im input gradient"""
        def build():
            n, cin, kh, kw = self.weight.shape
            cin_p, n_p = ops.ceil_to(cin, 8), ops.ceil_to(n, 8)
            wk = torch.zeros(n_p, kh * kw, cin_p, dtype=torch.bfloat16, device=self.weight.device)
            wk[:n, :, :cin] = self.weight.detach().permute(0, 2, 3, 1).reshape(n, kh * kw, cin).to(torch.bfloat16)
            return ops.pack_linear_weight(wk.reshape(n_p, kh * kw * cin_p).t().contiguous(), None)
        return self._cached("c", (self.weight,), build)

    def forward(self, x, *, up2=False, rowvec=None, residual=None, act=ops.ACT_NONE):
        s = self.stride[0]
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            if act not in (ops.ACT_NONE, ops.ACT_RELU):
                raise NotImplementedError("fused SiLU / GEGLU are inference-only; apply the activation as its own op when training")
            from ..autograd import Conv2dFn
            relu = act == ops.ACT_RELU
            if self.asym_pad:
                return Conv2dFn.apply(x, self.weight, self.bias, residual, rowvec, self, s, (0, 0), False,
                                      (x.shape[1] // 2, x.shape[2] // 2), relu)
            return Conv2dFn.apply(x, self.weight, self.bias, residual, rowvec, self, s,
                                  (self.padding[0], self.padding[1]), up2, None, relu)
        if self.asym_pad:
            out_hw = (x.shape[1] // 2, x.shape[2] // 2)
            return ops.conv2d(x, self.packed(), stride=s, pad=(0, 0), out_hw=out_hw, rowvec=rowvec,
                              residual=residual, act=act, gn_groups=self.gn_groups)
        if (up2 and _UP2_SPLIT and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16 and rowvec is None and residual is None
                and s == 1 and self.kernel_size == (3, 3) and self.padding[0] == 1 and self.in_channels % 64 == 0):
            # inference, bf16: the upsampling convolution as four 2x2 convolutions of the low-resolution input
            return ops.conv2d_up2(x, self.packed_up2(), act=act, gn_groups=self.gn_groups)
        return ops.conv2d(x, self.packed(), stride=s, pad=(self.padding[0], self.padding[1]), up2=up2,
                          rowvec=rowvec, residual=residual, act=act, gn_groups=self.gn_groups)


class HipLinear(nn.Linear, _Cached):
    def packed(self) -> ops.PackedWeight:
        return self._cached("w", (self.weight, self.bias), lambda: ops.pack_linear_weight(self.weight, self.bias))

    def packed_geglu(self) -> ops.PackedWeight:
        return self._cached("g", (self.weight, self.bias), lambda: ops.pack_geglu_weight(self.weight, self.bias))

    def packed_t(self) -> ops.PackedWeight:
        """W^T as a packed [K, N] layer: dX = dY . W is the forward GEMM on it"""
        def build():
            w = self.weight.detach()
            n, k = w.shape
            if w.is_cuda and w.dtype == torch.bfloat16 and w.is_contiguous() and n % 8 == 0 and k % 8 == 0:
                return ops.pack_linear_weight(ops.transpose(w), None)       # one kernel, result aliases as the pack
            return ops.pack_linear_weight(w.t().contiguous(), None)
        return self._cached("t", (self.weight,), build)

    def forward(self, x, *, residual=None, rowvec=None, rows_per_image=0, act=ops.ACT_NONE, geglu=False, out=None):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            if act != ops.ACT_NONE or geglu or out is not None:
                raise NotImplementedError("fused activations / preallocated outputs are inference-only")
            from ..autograd import LinearFn
            return LinearFn.apply(x, self.weight, self.bias, residual, rowvec, self, rows_per_image)
        pw = self.packed_geglu() if geglu else self.packed()
        return ops.linear(x, pw, residual=residual, rowvec=rowvec, rows_per_image=rows_per_image, act=act, out=out)


class HipGroupNorm(nn.GroupNorm, _Cached):
    def affine_f32(self):
        return self._cached("a", (self.weight, self.bias),
                            lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()))

    def forward(self, x, silu: bool = False, passthrough: bool = False):
        """``passthrough``: also return x for the residual link around the block -> (y, x).  On the training route that
        second output is a node of the norm's autograd Function, so that both gradients of x meet in ONE backward
        kernel (autograd.GroupNormFn); use it, not x, as the residual operand."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            from ..autograd import GRAD_FUSION, GroupNormFn
            if passthrough and not GRAD_FUSION:
                return GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, silu, self, False), x
            return GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, silu, self, passthrough)
        g, b = self.affine_f32()
        y = ops.groupnorm(x, g, b, self.num_groups, self.eps, silu)
        return (y, x) if passthrough else y


class HipLayerNorm(nn.LayerNorm, _Cached):
    def affine_f32(self):
        return self._cached("a", (self.weight, self.bias),
                            lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()))

    def forward(self, x, passthrough: bool = False):
        """``passthrough``: see HipGroupNorm.forward"""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            from ..autograd import GRAD_FUSION, LayerNormFn
            if passthrough and not GRAD_FUSION:
                return LayerNormFn.apply(x.contiguous(), self.weight, self.bias, self.eps, self, False), x
            return LayerNormFn.apply(x.contiguous(), self.weight, self.bias, self.eps, self, passthrough)
        g, b = self.affine_f32()
        y = ops.layernorm(x, g, b, self.eps)
        return (y, x) if passthrough else y


def training(*tensors_or_params) -> bool:
    """True when autograd is recording and any operand wants a gradient: modules then take the
    unfused, differentiable route (every op a torch.autograd.Function over the HIP kernels)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors_or_params)


def fused_rows(module: _Cached, name: str, layers, bias: bool):
    """Packs several Linear layers that share an input into one [sum N, K] weight."""
    params = []
    for l in layers:
        params += [l.weight, l.bias]

    def build():
        w = torch.cat([l.weight.detach() for l in layers], 0)
        b = None
        if bias:
            b = torch.cat([l.bias.detach() if l.bias is not None else torch.zeros(l.out_features, device=w.device, dtype=w.dtype)
                           for l in layers], 0)
        return ops.pack_linear_weight(w, b)

    return module._cached(name, params, build)
