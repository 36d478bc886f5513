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


def _cast_pair(dg, db, dt):
    """(dg, db) are rows 0 / 1 of one f32 [2, C] tensor: one conversion kernel for both"""
    if dt == torch.float32:
        return dg, db
    both = dg._base.to(dt) if dg._base is not None and dg._base is db._base else None
    return (both[0], both[1]) if both is not None else (dg.to(dt), db.to(dt))


def _grad_sink(param):
    """The trainer's flat-gradient sink of ``param`` when one is accepting direct writes, else None.
    With a sink, a weight gradient is computed straight into the parameter's slot of the flat gradient
    buffer (``sink.write``) and the Function returns None for it: no temporary, no accumulation kernel."""
    ent = getattr(param, "_lbmk_grad_sink", None)
    return ent if ent is not None and ent.active else None


# LBM_NO_GRAD_FUSION=1: bias / norm gradients go through autograd's accumulation and residual-link gradients are added by
# autograd (same-box A/B runs)
import os
GRAD_FUSION = os.environ.get("LBM_NO_GRAD_FUSION") is None


# bytes a transposed operand of the convolution weight gradient may span (32-bit byte offsets in the GEMM kernel); tests
# lower it to exercise the grouping
WGRAD_WINDOW_BYTES = (1 << 31) - 1


def _param_slots(*params):
    """(sink, slot views) when the gradients of these (small) parameters can be written straight into the trainer's flat
    gradient buffer by the kernel that produces them, else (None, None)"""
    if not GRAD_FUSION or any(p is None for p in params):
        return None, None
    sink = _grad_sink(params[0])
    if sink is None or any(_grad_sink(p) is not sink for p in params[1:]):
        return None, None
    views = sink.slots(list(params))
    return (sink, views) if views is not None else (None, None)


def _linear_maybe_splitk(x, pw, residual):
    """the few-row, deep-K linear layers of a training step as a split-K launch (ops.splitk_chunks); None = not eligible"""
    if x.dtype != bf16 or pw.geglu or pw.KH != 1 or pw.w.dtype != bf16 or pw.N % 8:
        return None
    x2 = _rows(x)
    if x2.stride(1) != 1 or x2.shape[1] != pw.K:
        return None
    S = ops.splitk_chunks(x2.shape[0], pw.N, pw.K)
    if not S:
        return None
    r2 = None
    if residual is not None:
        r2 = _rows(residual)
        if r2.dtype != bf16 or r2.stride(1) != 1 or r2.shape != (x2.shape[0], pw.N):
            return None
    return ops.linear_splitk(x2, pw, S, r2).reshape(*x.shape[:-1], pw.N)


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (+ residual) (+ rowvec[image]); ``mod`` supplies the cached packed weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rowvec, mod, rows_per_image):
        y = _linear_maybe_splitk(x, mod.packed(), residual) if rowvec is None else None
        if y is None:
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
            if ops.gemm_nn_ok(dy2, weight) and ops.gemm_nn_wins(dy2.shape[0], N, K):
                dx = ops.gemm_nn(dy2, weight)                   # reads the parameter as stored: no W^T copy per step
            else:
                dx = _linear_maybe_splitk(dy2, ctx.mod.packed_t(), None)
                if dx is None:
                    dx = ops.linear(dy2, ctx.mod.packed_t())
            dx = dx[:, :K].reshape(x.shape)
        if ctx.needs_input_grad[1]:
            xc = x2.contiguous() if x2.stride(1) != 1 else x2
            sink = _grad_sink(ctx.mod.weight)
            if sink is not None and dy2.shape[1] == N and xc.shape[1] == K:
                sink.write([ctx.mod.weight], lambda out: ops.gemm_tn(dy2, xc, out=out), reads=(dy2, xc))
            else:
                dw = ops.gemm_tn(dy2, xc, out_dtype=weight.dtype)[:N, :K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sink, views = _param_slots(ctx.mod.bias) if dy2.shape[1] == N else (None, None)
            if sink is not None:
                ops.colsum(dy2, out=views[0])
                sink.commit([ctx.mod.bias])
            else:
                db = ops.colsum(dy2, out_dtype=weight.dtype)[0]
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        if ctx.has_rv and ctx.needs_input_grad[4]:
            drv = ops.colsum(dy2, ctx.rows_per_image, out_dtype=bf16)
        return dx, dw, db, dres, drv, None, None


class Conv2dFn(torch.autograd.Function):
    """NHWC convolution (3x3 / 1x1, stride 1 or 2, optional fused nearest x2 upsample, asymmetric pad)
    with bias, per-image row vector (time embedding) and residual epilogues."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rowvec, mod, stride, pad, up2, out_hw, relu=False):
        y = ops.conv2d(x, mod.packed(), stride=stride, pad=pad, up2=up2, out_hw=out_hw, rowvec=rowvec, residual=residual,
                       act=ops.ACT_RELU if relu else ops.ACT_NONE)
        ctx.save_for_backward(x, weight, *((y,) if relu else ()))
        ctx.mod, ctx.geom, ctx.relu = mod, (stride, pad, up2, (y.shape[1], y.shape[2])), relu
        ctx.has_bias, ctx.has_res, ctx.has_rv = bias is not None, residual is not None, rowvec is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors[:2]
        stride, pad, up2, (OH, OW) = ctx.geom
        N, Cin, KH, KW = weight.shape
        dy = dy.contiguous()
        if ctx.relu:
            dy = ops.relu_bwd(ctx.saved_tensors[2], dy)
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
            def wgrad():
                # the transposed operands are [rows][M] matrices addressed with 32-bit byte offsets: beyond 2 GiB (8 images
                # of 128 x 128 latent pixels times 8640 im2col rows) the images are processed in groups, partial sums in f32
                K9, per = KH * KW * cin_p, OH * OW
                max_cols = WGRAD_WINDOW_BYTES // (2 * max(K9, npad)) // 8 * 8
                nb = min(B, max_cols // per)
                if nb < 1:
                    raise NotImplementedError(f"convolution weight gradient: one image's {per} output pixels x {K9} im2col rows "
                                              "exceed the 2 GiB operand window")
                dwk = None
                for b0 in range(0, B, nb):
                    b1 = min(B, b0 + nb)
                    col_t = ops.im2col_t(x[b0:b1], cin_p, KH, KW, stride, pad, up2, (OH, OW))       # [KH*KW*cin_p, rows of the group]
                    part = ops.gemm_tn_t(ops.transpose(dy2[b0 * per:b1 * per]), col_t, out_dtype=torch.float32)   # [npad, KH*KW*cin_p]
                    dwk = part if dwk is None else dwk.add_(part)
                return dwk[:N].view(N, KH * KW, cin_p)[:, :, :Cin].permute(0, 2, 1)       # [N, Cin, KH*KW] strided view
            sink = _grad_sink(ctx.mod.weight)
            if sink is not None:
                sink.write([ctx.mod.weight], lambda out: out.view(N, Cin, KH * KW).copy_(wgrad()), reads=(x, dy2))
            else:
                dw = wgrad().reshape(N, Cin, KH, KW).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sink, views = _param_slots(ctx.mod.bias) if npad == N else (None, None)
            if sink is not None:
                ops.colsum(dy2, out=views[0])
                sink.commit([ctx.mod.bias])
            else:
                db = ops.colsum(dy2, out_dtype=weight.dtype)[0, :N]
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        if ctx.has_rv and ctx.needs_input_grad[4]:
            drv = ops.colsum(dy2, OH * OW, out_dtype=bf16)
        return dx, dw, db, dres, drv, None, None, None, None, None, None


class GroupNormFn(torch.autograd.Function):
    """GroupNorm with optional fused SiLU; gamma / beta are the module's parameters (any float dtype).

    ``passthrough``: the Function also returns its input (as a second output) for the tensor's OTHER consumer, the
    residual link around the block (x -> norm -> ... -> + x).  Both gradients of x then arrive at this node, and the
    backward kernel adds the residual's to dx in its own pass, where autograd would launch an elementwise addition.
    ``mod``: the module that owns gamma / beta (their gradients go straight into the trainer's flat gradient buffer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu, mod=None, passthrough=False):
        g32, b32 = ops.f32_of(weight), ops.f32_of(bias)
        y, stats = ops.groupnorm_train(x, g32, b32, groups, eps, silu)
        ctx.save_for_backward(x, stats, g32, b32)
        ctx.cfg = (groups, silu, weight.dtype)
        ctx.mod, ctx.passthrough = mod, passthrough
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, stats, g32, b32 = ctx.saved_tensors
        groups, silu, wdt = ctx.cfg
        sink, views = (None, None)
        if ctx.mod is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            sink, views = _param_slots(ctx.mod.weight, ctx.mod.bias)
        add = dres.contiguous() if dres is not None else None
        if sink is not None:
            dx = ops.groupnorm_bwd(x, dy.contiguous(), stats, g32, b32, groups, silu, add=add, dg_out=views[0], db_out=views[1])[0]
            sink.commit([ctx.mod.weight, ctx.mod.bias])
            return (dx, None, None, None, None, None, None, None)
        dx, dg, db = ops.groupnorm_bwd(x, dy.contiguous(), stats, g32, b32, groups, silu, add=add)
        return (dx, *_cast_pair(dg, db, wdt), None, None, None, None, None)


class LayerNormFn(torch.autograd.Function):
    """``mod`` / ``passthrough``: see GroupNormFn (x -> LayerNorm -> attention / feed-forward -> + x)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, mod=None, passthrough=False):
        g32, b32 = ops.f32_of(weight), ops.f32_of(bias)
        y = ops.layernorm(x, g32, b32, eps)
        ctx.save_for_backward(x, g32)
        ctx.cfg = (eps, weight.dtype)
        ctx.mod, ctx.passthrough = mod, passthrough
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, g32 = ctx.saved_tensors
        eps, wdt = ctx.cfg
        sink, views = (None, None)
        if ctx.mod is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            sink, views = _param_slots(ctx.mod.weight, ctx.mod.bias)
        add = dres.contiguous() if dres is not None else None
        if sink is not None:
            dx = ops.layernorm_bwd(x, dy.contiguous(), g32, eps, add=add, dg_out=views[0], db_out=views[1])[0]
            sink.commit([ctx.mod.weight, ctx.mod.bias])
            return (dx, None, None, None, None, None)
        dx, dg, db = ops.layernorm_bwd(x, dy.contiguous(), g32, eps, add=add)
        return (dx, *_cast_pair(dg, db, wdt), None, None, None)


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, dy.contiguous())


class GegluFn(torch.autograd.Function):
    """h = [value | gate] -> value * gelu(gate) (diffusers GEGLU)."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return ops.geglu_fwd(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return ops.geglu_bwd(h, dy.contiguous())


def _heads(t: torch.Tensor, H: int, D: int) -> torch.Tensor:
    """[B, N, H*D] -> strided view [B, H, N, D] (no copy)"""
    B, N, _ = t.shape
    return t.view(B, N, H, D).permute(0, 2, 1, 3)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) v per head, heads interleaved along the channel axis ([B, N, H*D]).
    Forward: the flash kernel.  Backward, head dims 40 / 64 / 80 / 160 with token counts that are multiples of 64: the fused
    flash backward kernels (scores never materialised).  Otherwise probabilities are recomputed (q, k, v are
    all that is saved) and every product is one implicit-GEMM launch batched over (image, head):
      dV = P^T dO,  dP = dO V^T,  dS = scale * P * (dP - rowsum(dP * P)),  dQ = dS K,  dK = dS^T Q."""

    @staticmethod
    def forward(ctx, q, k, v, heads, dim_head, scale):
        ctx.cfg = (heads, dim_head, scale)
        Nk = k.shape[1]
        ctx.nk = Nk
        ctx.nq = q.shape[1]
        if ctx.nq % 8:
            # a 2x2 / 6x6 token grid (deepest level of a small training resolution): zero query rows up to a multiple
            # of 8; their outputs are dropped and their output gradients are zero, so they reach no gradient
            q = torch.nn.functional.pad(q, (0, 0, 0, -ctx.nq % 8))
        if Nk % 8:
            # a handful of context tokens (lighting parameters): zero rows up to a multiple of 8, masked out of the softmax
            k = torch.nn.functional.pad(k, (0, 0, 0, -Nk % 8))
            v = torch.nn.functional.pad(v, (0, 0, 0, -Nk % 8))
        elif (ctx.nq == q.shape[1] and ops.flash_attn_bwd_supported(dim_head, q.shape[1], Nk) and q.is_contiguous()
                and k.is_contiguous() and v.is_contiguous()):
            # fused backward available: keep the output and the row log-sum-exp instead of recomputing P
            o, lse = ops.flash_attn(q, k, ops.transpose(v), heads, dim_head, scale, return_lse=True)
            ctx.save_for_backward(q, k, v, o, lse)
            ctx.fused = True
            return o
        ctx.fused = False
        if dim_head in (40, 64, 80, 160):
            o = ops.flash_attn(q, k[:, :Nk], ops.transpose(v), heads, dim_head, scale)     # keys past Nk are masked
        else:
            # head dims without a flash kernel (VAE mid block: one head of 512): materialised scores
            B, Nq, C = q.shape
            p = ops.softmax_rows(AttentionFn._scores(q, k, heads, dim_head, scale, Nk))
            o = torch.empty_like(q)
            ops.gemm_nt(p, ops.transpose(v).view(B, heads, dim_head, k.shape[1]), out=_heads(o, heads, dim_head))
        ctx.save_for_backward(q, k, v)
        return o if ctx.nq == q.shape[1] else o[:, :ctx.nq].contiguous()

    @staticmethod
    def _scores(q, k, H, D, scale, nk):
        s = ops.gemm_nt(_heads(q, H, D), _heads(k, H, D), out_dtype=torch.float32, alpha=scale)
        if nk < k.shape[1]:
            s[..., nk:] = float("-inf")
        return s

    @staticmethod
    def backward(ctx, do):
        H, D, scale = ctx.cfg
        do = do.contiguous()
        if ctx.fused:
            q, k, v, o, lse = ctx.saved_tensors
            return (*ops.flash_attn_bwd(q, k, v, o, do, lse, H, D, scale), None, None, None)
        q, k, v = ctx.saved_tensors
        B, Nq, C = q.shape
        Nk = k.shape[1]
        if ctx.nq < Nq:
            do = torch.nn.functional.pad(do, (0, 0, 0, Nq - ctx.nq))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        q4, k4, v4, do4 = (_heads(t, H, D) for t in (q, k, v, do))
        p = ops.softmax_rows(AttentionFn._scores(q, k, H, D, scale, ctx.nk))                 # [B, H, Nq, Nk] bf16
        # channel-major copies: [B, C, N] read as [B, H, D, N]
        qt4 = ops.transpose(q).view(B, H, D, Nq)
        kt4 = ops.transpose(k).view(B, H, D, Nk)
        dot4 = ops.transpose(do).view(B, H, D, Nq)
        pt = ops.transpose(p.view(B * H, Nq, Nk)).view(B, H, Nk, Nq)
        ops.gemm_nt(pt, dot4, out=_heads(dv, H, D))                                          # dV = P^T dO
        dp = ops.gemm_nt(do4, v4, out_dtype=torch.float32)                                   # dP = dO V^T
        ds = ops.softmax_bwd_rows(p, dp, scale)
        ops.gemm_nt(ds, kt4, out=_heads(dq, H, D))                                           # dQ = dS K
        dst = ops.transpose(ds.view(B * H, Nq, Nk)).view(B, H, Nk, Nq)
        ops.gemm_nt(dst, qt4, out=_heads(dk, H, D))                                          # dK = dS^T Q
        return dq[:, :ctx.nq], dk[:, :ctx.nk], dv[:, :ctx.nk], None, None, None


class SelfAttnQKVFn(torch.autograd.Function):
    """Self-attention whose three bias-free input projections run as ONE GEMM over the stacked weights:
    qkv = x @ [W_a; W_b; W_c]^T, o = softmax(q k^T * scale) v.  ``mod`` (the attention module) supplies the
    stacked packed weight, its transpose and the column order of (q, k, v) in the stack.  Backward: the fused
    flash kernels write dq / dk / dv into one [B, N, 3C] buffer, so the input gradient and the weight
    gradient are one GEMM each as well (instead of three plus two accumulations)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, mod, heads, dim_head, scale):
        C = heads * dim_head
        pw, cols = mod.packed_qkv()
        qkv = ops.linear(x, pw)                                                   # [B, N, 3C]
        q, k, v = (qkv[..., cols[r]:cols[r] + C] for r in "qkv")
        o, lse = ops.flash_attn(q, k, ops.transpose(v), heads, dim_head, scale, return_lse=True)
        ctx.save_for_backward(x, qkv, o, lse)
        ctx.mod, ctx.cfg, ctx.cols, ctx.wdt = mod, (heads, dim_head, scale), cols, wq.dtype
        return o

    @staticmethod
    def backward(ctx, do):
        x, qkv, o, lse = ctx.saved_tensors
        H, D, scale = ctx.cfg
        C, cols = H * D, ctx.cols
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        qkvt = ops.transpose(qkv)                                                 # [B, 3C, N]
        q, k, v = (qkv[..., cols[r]:cols[r] + C] for r in "qkv")
        dq, dk, dv = (dqkv[..., cols[r]:cols[r] + C] for r in "qkv")
        ops.flash_attn_bwd(q, k, v, o, do, lse, H, D, scale, out=(dq, dk, dv),
                           transposed=(qkvt[:, cols["q"]:cols["q"] + C], qkvt[:, cols["k"]:cols["k"] + C]))
        d2, x2 = _rows(dqkv), _rows(x)
        dx = None
        if ctx.needs_input_grad[0]:
            stack = ctx.mod._qkv_stack()[0]
            is_view = stack.data_ptr() in (mod_w.data_ptr() for mod_w in (ctx.mod.to_q.weight, ctx.mod.to_k.weight, ctx.mod.to_v.weight))
            if is_view and ops.gemm_nn_ok(d2, stack) and ops.gemm_nn_wins(d2.shape[0], *stack.shape):
                # (a stack that had to be concatenated is no cheaper than W^T)
                dx = ops.gemm_nn(d2, stack).reshape(x.shape)
            else:
                dx = ops.linear(d2, ctx.mod.packed_qkv_t())[:, :x.shape[-1]].reshape(x.shape)
        dws = [None, None, None]
        mod = ctx.mod
        ws = [mod.to_q.weight, mod.to_k.weight, mod.to_v.weight]
        sink = _grad_sink(ws[0])
        if all(ctx.needs_input_grad[1:4]) and sink is not None and sink.stacked(ws, mod._qkv_stack()[0]):
            # the three weights are one [3C, K] block of the flat buffer, and so are their gradient slots
            xc = x2.contiguous() if x2.stride(1) != 1 else x2
            sink.write(ws, lambda out: ops.gemm_tn(d2, xc, out=out), reads=(d2, xc))
        elif any(ctx.needs_input_grad[1:4]):
            dw = ops.gemm_tn(d2, x2.contiguous() if x2.stride(1) != 1 else x2, out_dtype=ctx.wdt)   # [3C, K]
            dws = [dw[cols[r]:cols[r] + C, :x.shape[-1]] if ctx.needs_input_grad[1 + i] else None
                   for i, r in enumerate("qkv")]
        return (dx, *dws, None, None, None, None)


class MaxPool2Fn(torch.autograd.Function):
    """2x2 / stride-2 max pooling on NHWC"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, dy.contiguous())


class LpipsPrepFn(torch.autograd.Function):
    """LPIPS input stage: optional clamp to [-1, 1], valid-pixel mask, ScalingLayer, NCHW -> NHWC bf16."""

    @staticmethod
    def forward(ctx, x, mask, shift, scale, clamp):
        x = x.contiguous()
        ctx.save_for_backward(x, mask if mask is not None else torch.empty(0))
        ctx.cfg = (mask is not None, tuple(scale), clamp)
        return ops.lpips_prep(x, mask, shift, scale, clamp)

    @staticmethod
    def backward(ctx, dy):
        x, mask = ctx.saved_tensors
        has_mask, scale, clamp = ctx.cfg
        return ops.lpips_prep_bwd(x, mask if has_mask else None, dy.contiguous(), scale, clamp), None, None, None, None


class LpipsHeadFn(torch.autograd.Function):
    """one LPIPS layer: unit-normalise both feature maps over channels, squared difference, 1x1 linear head
    with non-trainable weights here (gradient to the first feature map only), spatial mean -> [B]."""

    @staticmethod
    def forward(ctx, f0, f1, w, keep):
        ctx.save_for_backward(f0, f1, w, keep if keep is not None else torch.empty(0))
        ctx.has_keep = keep is not None
        return ops.lpips_head(f0, f1, w, keep)

    @staticmethod
    def backward(ctx, gout):
        f0, f1, w, keep = ctx.saved_tensors
        return ops.lpips_head_bwd(f0, f1, w, keep if ctx.has_keep else None, gout), None, None, None


class NchwToNhwcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.cfg = (x.shape[1], x.dtype)
        return ops.nchw_to_nhwc(x)

    @staticmethod
    def backward(ctx, dy):
        C, dt = ctx.cfg
        return ops.nhwc_to_nchw(dy.contiguous(), C, out_dtype=dt)


class NhwcToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C, out_dtype):
        ctx.cp = x.shape[-1]
        return ops.nhwc_to_nchw(x, C, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        dx = ops.nchw_to_nhwc(dy.contiguous())           # channels zero-padded to a multiple of 8
        if dx.shape[-1] != ctx.cp:
            dx = torch.nn.functional.pad(dx, (0, ctx.cp - dx.shape[-1]))
        return dx, None, None


class LatentLossFn(torch.autograd.Function):
    """per-sample mean of ((pred - (z_src - z)) * mask)^2 or |.| (src/lbm/models/lbm/lbm_model.py:186-195)."""

    @staticmethod
    def forward(ctx, pred, z_src, z, mask, l1):
        ctx.save_for_backward(pred, z_src, z, mask if mask is not None else torch.empty(0))
        ctx.l1, ctx.has_mask = l1, mask is not None
        return ops.latent_loss(pred.contiguous(), z_src.contiguous(), z.contiguous(), mask, l1=l1)

    @staticmethod
    def backward(ctx, gout):
        pred, z_src, z, mask = ctx.saved_tensors
        return ops.latent_loss_bwd(pred, z_src, z, mask if ctx.has_mask else None, gout, ctx.l1), None, None, None, None


class PixelLossFn(torch.autograd.Function):
    """per-sample mean of ((c(dec) - target) * mask)^2 or |.| (lbm_model.py:197-249, l2 / l1); c clips to [-1, 1]
    when ``clamp`` (a VAE-decoded prediction, :233-234) and is the identity otherwise (no VAE, :235-236)."""

    @staticmethod
    def forward(ctx, dec, target, mask, l1, clamp=True):
        dec, target = dec.contiguous(), target.contiguous()
        ctx.save_for_backward(dec, target, mask if mask is not None else torch.empty(0))
        ctx.l1, ctx.has_mask, ctx.clamp = l1, mask is not None, clamp
        return ops.pixel_loss(dec, target, mask, l1=l1, clamp=clamp)

    @staticmethod
    def backward(ctx, gout):
        dec, target, mask = ctx.saved_tensors
        return (ops.pixel_loss_bwd(dec, target, mask if ctx.has_mask else None, gout, ctx.l1, ctx.clamp), None, None,
                None, None)


class AxpyRowsFn(torch.autograd.Function):
    """x[b] + a[b] * y[b] in f32, differentiable in y (predicted x0 = x_t - sigma * prediction)."""

    @staticmethod
    def forward(ctx, x, y, a):
        ctx.save_for_backward(a)
        ctx.ydt = y.dtype
        return ops.axpy_rows(x.contiguous(), y.contiguous(), a)

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return None, ops.scale_rows(g.contiguous(), a).to(ctx.ydt), None


class ScaleShiftFn(torch.autograd.Function):
    """y = x * a + b elementwise on bf16 (latent un-scaling ahead of the VAE decoder)."""

    @staticmethod
    def forward(ctx, x, a, b):
        ctx.a = a
        return ops.scale_shift(x, a, b)

    @staticmethod
    def backward(ctx, g):
        return ops.scale_shift(g.contiguous(), ctx.a, 0.0), None, None
