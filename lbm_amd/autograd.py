"""torch.autograd.Function wrappers: forward and backward both run on the HIP kernels.

The reference trains through torch autograd over diffusers modules (Lightning ``training_step`` around
``LBMModel.forward``, src/lbm/models/lbm/lbm_model.py:96-184).  Here autograd only records the graph
and routes gradients; every gradient is computed by liblbmk:
  dX = dY . W        the forward implicit GEMM on a transposed (convolutions: flipped) packed weight
  dW = dY^T . col(X) implicit GEMM over the transposed operands, split across the row range
  db = column sums of dY
"""
from typing import Optional

import torch

from . import ops

bf16 = torch.bfloat16


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (+ residual) (+ rowvec[image]); ``mod`` supplies the cached packed weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rowvec, mod, rows_per_image):
        y = ops.linear(x, mod.packed(), residual=residual, rowvec=rowvec, rows_per_image=rows_per_image)
        ctx.save_for_backward(x, weight)
        ctx.mod, ctx.has_bias, ctx.rows_per_image = mod, bias is not None, rows_per_image
        ctx.has_res, ctx.has_rv = residual is not None, rowvec is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dy2, x2 = _rows(dy), _rows(x)
        N, K = weight.shape
        dx = dw = db = dres = drv = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy2, ctx.mod.packed_t())[:, :K].reshape(x.shape)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dy2, x2.contiguous() if x2.stride(1) != 1 else x2, out_dtype=torch.float32)[:N, :K].to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy2)[0].to(weight.dtype)
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        if ctx.has_rv and ctx.needs_input_grad[4]:
            drv = ops.colsum(dy2, ctx.rows_per_image, out_dtype=bf16)
        return dx, dw, db, dres, drv, None, None


class Conv2dFn(torch.autograd.Function):
    """NHWC convolution (3x3 / 1x1, stride 1 or 2, optional fused nearest x2 upsample, asymmetric pad)
    with bias, per-image row vector (time embedding) and residual epilogues."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rowvec, mod, stride, pad, up2, out_hw):
        y = ops.conv2d(x, mod.packed(), stride=stride, pad=pad, up2=up2, out_hw=out_hw, rowvec=rowvec, residual=residual)
        ctx.save_for_backward(x, weight)
        ctx.mod, ctx.geom = mod, (stride, pad, up2, (y.shape[1], y.shape[2]))
        ctx.has_bias, ctx.has_res, ctx.has_rv = bias is not None, residual is not None, rowvec is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, up2, (OH, OW) = ctx.geom
        N, Cin, KH, KW = weight.shape
        dy = dy.contiguous()
        B, IH, IW, Cs = x.shape
        dy2 = _rows(dy)
        npad = dy.shape[-1]          # output channels padded to 8
        cin_p = ops.ceil_to(Cin, 8)
        dx = dw = db = dres = drv = None
        if ctx.needs_input_grad[0]:
            if stride == 1 and not up2 and pad == ((KH - 1) // 2, (KW - 1) // 2):
                # full correlation with the flipped, transposed filter
                dx = ops.conv2d(dy, ctx.mod.packed_dgrad(), stride=1, pad=(KH - 1 - pad[0], KW - 1 - pad[1]))
            else:
                dcol = ops.linear(dy2, ctx.mod.packed_col_t())           # [M, KH*KW*cin_p]
                dx = ops.col2im(dcol, B, IH, IW, OH, OW, cin_p, KH, KW, stride, pad, up2)
            if dx.shape[-1] != Cs:
                dx = dx[..., :Cs].contiguous() if dx.shape[-1] > Cs else torch.nn.functional.pad(dx, (0, Cs - dx.shape[-1]))
        if ctx.needs_input_grad[1]:
            col_t = ops.im2col_t(x, cin_p, KH, KW, stride, pad, up2, (OH, OW))     # [KH*KW*cin_p, M]
            dwk = ops.gemm_tn_t(ops.transpose(dy2), col_t, out_dtype=torch.float32)   # [npad, KH*KW*cin_p]
            dw = dwk[:N].reshape(N, KH * KW, cin_p)[:, :, :Cin].permute(0, 2, 1).reshape(N, Cin, KH, KW).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy2)[0, :N].to(weight.dtype)
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        if ctx.has_rv and ctx.needs_input_grad[4]:
            drv = ops.colsum(dy2, OH * OW, out_dtype=bf16)
        return dx, dw, db, dres, drv, None, None, None, None, None
