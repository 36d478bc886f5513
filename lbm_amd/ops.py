"""Tensor-level wrappers over the C-ABI kernels.

PyTorch is used for device memory and streams only: every function here checks its
operands, allocates the output with ``torch.empty`` and hands raw pointers plus the
current HIP stream to liblbmk.  Activations are bf16 channels-last: a ``[B, H, W, C]``
tensor (contiguous) which is at the same time the ``[B*H*W, C]`` token matrix.
"""
import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _C
from ._C import ACT_GEGLU, ACT_NONE, ACT_RELU, ACT_SILU, DT_BF16, DT_F32, IgemmDesc, check, lib

bf16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lbm_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise TypeError(f"unsupported dtype {t.dtype} (bf16 / f32 only)")


def _act_f32(x: torch.Tensor, w: torch.Tensor) -> bool:
    """fp32 mode when both the activation and the weight operand are f32; bf16 otherwise (mixing is an error)"""
    if x.dtype == torch.float32 and w.dtype == torch.float32:
        return True
    if x.dtype == bf16 and w.dtype == bf16:
        return False
    raise TypeError(f"activation {x.dtype} with weights {w.dtype}: a module runs in bf16 or in f32 (fp32 mode), "
                    "cast the model with .to(dtype)")


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------
# weight packing (host-side, done once per module)
# --------------------------------------------------------------------------------------
@dataclass
class PackedWeight:
    w: torch.Tensor                 # bf16 [N, Kp]
    bias: Optional[torch.Tensor]    # f32 [N] (packed order) or None
    N: int
    K: int
    Kp: int
    Cin: int                        # channels per tap after padding to a multiple of 8
    KH: int
    KW: int
    geglu: bool = False
    alg_nk: int = 0                 # un-padded N*K of the layer (algorithmic MACs per output row)
    k_order: int = 0                # 1: K runs chunk-major (64 channels x all taps), see lbmk_igemm_desc.k_order


def f32_of(p: torch.Tensor) -> torch.Tensor:
    """contiguous f32 copy of a (bias / norm) parameter.  The trainer keeps f32 shadows of all 1-D parameters, refreshed
    by one multi-tensor copy after each optimizer step (HipAdamW.invalidate) and stamped with the parameter's version;
    a current shadow is returned as is, anything else is converted here."""
    ent = getattr(p, "_lbmk_f32", None)
    if ent is not None and ent[1] == p._version and ent[0].device == p.device:
        return ent[0]
    return p.detach().float().contiguous()


def pack_conv_weight(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedWeight:
    """[Cout, Cin, KH, KW] -> bf16 [Cout_p, Kp] with k = (kh*KW + kw)*Cin_p + c.
    Both channel counts are padded to multiples of 8 with zeros, so every channels-last
    tensor in the pipeline has 16-byte pixel rows (a 3- or 4-channel output carries zero
    channels up to 8)."""
    N, Cin, KH, KW = weight.shape
    cin_p = ceil_to(Cin, 8)
    n_p = ceil_to(N, 8)
    K = KH * KW * cin_p
    Kp = ceil_to(K, 64)
    if weight.dtype == torch.float32:
        # fp32 mode: f32 operand in tap-major K order (the f32 kernel takes any geometry)
        w = torch.zeros(n_p, KH * KW, cin_p, dtype=torch.float32, device=weight.device)
        w[:N, :, :Cin] = weight.detach().permute(0, 2, 3, 1).reshape(N, KH * KW, Cin)
        b = None
        if bias is not None:
            b = torch.zeros(n_p, dtype=torch.float32, device=weight.device)
            b[:N] = bias.detach().float()
        return PackedWeight(w.reshape(n_p, K).contiguous(), b, n_p, K, K, cin_p, KH, KW, alg_nk=N * KH * KW * Cin)
    if weight.is_cuda and weight.dtype == bf16 and weight.is_contiguous():
        return _pack_conv_weight_dev(weight.detach(), bias, False)
    w = torch.zeros(N, KH * KW, cin_p, dtype=bf16, device=weight.device)
    w[:, :, :Cin] = weight.detach().permute(0, 2, 3, 1).reshape(N, KH * KW, Cin).to(bf16)
    # 3x3 layers with whole 64-channel chunks: make the 9 taps of one chunk consecutive K tiles, so
    # successive tiles re-read nearly the same (one-pixel shifted) input lines
    k_order = 1 if (KH * KW > 1 and cin_p % 64 == 0) else 0
    if k_order:
        w = w.reshape(N, KH * KW, cin_p // 64, 64).permute(0, 2, 1, 3)
    packed = torch.zeros(n_p, Kp, dtype=bf16, device=weight.device)
    packed[:N, :K] = w.reshape(N, K)
    b = None
    if bias is not None:
        b = torch.zeros(n_p, dtype=torch.float32, device=weight.device)
        b[:N] = bias.detach().float()
    return PackedWeight(packed.contiguous(), b, n_p, K, Kp, cin_p, KH, KW, alg_nk=N * KH * KW * Cin, k_order=k_order)


def _pack_conv_weight_dev(weight: torch.Tensor, bias: Optional[torch.Tensor], dgrad: bool) -> PackedWeight:
    """one kernel: the forward operand, or (``dgrad``) the input-gradient filter (channels swapped, taps mirrored)"""
    N, Cin, KH, KW = weight.shape
    rows, cols = (Cin, N) if dgrad else (N, Cin)
    cols_p, rows_p = ceil_to(cols, 8), ceil_to(rows, 8)
    K = KH * KW * cols_p
    Kp = ceil_to(K, 64)
    k_order = 1 if (KH * KW > 1 and cols_p % 64 == 0) else 0
    packed = torch.empty(rows_p, Kp, dtype=bf16, device=weight.device)
    check(lib.lbmk_pack_conv_weight(weight.data_ptr(), packed.data_ptr(), N, Cin, KH, KW, rows_p, Kp, k_order,
                                    1 if dgrad else 0, _stream()), "lbmk_pack_conv_weight")
    b = None
    if bias is not None:
        if rows_p == rows:
            b = f32_of(bias)
        else:
            b = torch.zeros(rows_p, dtype=torch.float32, device=weight.device)
            b[:rows] = bias.detach().float()
    return PackedWeight(packed, b, rows_p, K, Kp, cols_p, KH, KW, alg_nk=rows * KH * KW * cols, k_order=k_order)


def up2_parity_filters(weight: torch.Tensor):
    """[N, Cin, 3, 3] -> [(w2 [N, Cin, 2, 2] f32, pad_t, pad_l)] in parity order 2 * py + px, such that
    conv3x3(pad 1)(upsample_nearest_2x(x))[:, :, py::2, px::2] == conv2x2(pad top/left pad_t/pad_l, bottom/right
    1 - pad)(x).  Output row 2y + py reads upsampled rows 2y + py - 1 ... + 1, i.e. source rows {y - 1, y, y} (py = 0) or
    {y, y, y + 1} (py = 1): the taps that land on one source pixel are summed (in f32)."""
    N, Cin, KH, KW = weight.shape
    assert KH == 3 and KW == 3
    w = weight.detach().float()
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}      # parity -> filter rows summed into tap 0 / tap 1
    parts = []
    for py in (0, 1):
        for px in (0, 1):
            w2 = torch.zeros(N, Cin, 2, 2, dtype=torch.float32, device=w.device)
            for dy in (0, 1):
                for dx in (0, 1):
                    w2[:, :, dy, dx] = w[:, :, rows[py][dy], :][:, :, :, rows[px][dx]].sum(dim=(2, 3))
            parts.append((w2, 1 - py, 1 - px))
    return parts


def pack_conv_weight_up2(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """Nearest-x2 upsample followed by a 3x3 / pad-1 convolution, as four 2x2 convolutions of the low-resolution
    input, one per output parity (``up2_parity_filters``; each summed tap is rounded to bf16 once).  Returns
    [(PackedWeight, pad_t, pad_l)] in parity order 2 * py + px; each part does 4/9 of the products of its quarter of
    the output."""
    return [(pack_conv_weight(w2.to(bf16).contiguous(), bias), pt, pl) for w2, pt, pl in up2_parity_filters(weight)]


def conv2d_up2(x: torch.Tensor, parts, *, act: int = ACT_NONE, tile: int = -1, gn_groups: int = 0) -> torch.Tensor:
    """x: bf16 [B, H, W, C] -> [B, 2H, 2W, N]: upsample + 3x3 convolution from ``pack_conv_weight_up2`` parts.  Each
    parity launch scatters its pixels into the shared output (lbmk_igemm_desc.out_up2) and, when ``gn_groups``, its
    statistics strips into one partial-sum buffer for the GroupNorm that follows."""
    pw0 = parts[0][0]
    _need_gpu(x, pw0.w)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == bf16 and not pw0.geglu
    B, IH, IW, Cs = x.shape
    out = torch.empty(B, 2 * IH, 2 * IW, pw0.N, dtype=bf16, device=x.device)
    partial = None
    if gn_groups:
        nfl = lib.lbmk_igemm_gn_partial_floats(4 * B * IH * IW, pw0.N, pw0.Cin, 2 * IH, 2 * IW, gn_groups)
        if nfl > 0 and (IH * IW) % 64 == 0:
            partial = torch.empty(nfl, dtype=torch.float32, device=x.device)
    for q, (pw, pad_t, pad_l) in enumerate(parts):
        d = IgemmDesc()
        d.a, d.w, d.out = x.data_ptr(), pw.w.data_ptr(), out.data_ptr()
        d.bias = _ptr(pw.bias)
        d.M, d.N, d.K, d.Kp = B * IH * IW, pw.N, pw.K, pw.Kp
        d.Cin, d.a_cs, d.a_coff = pw.Cin, Cs, 0
        d.IH, d.IW, d.OH, d.OW = IH, IW, IH, IW
        d.KH, d.KW, d.stride, d.pad_t, d.pad_l = 2, 2, 1, pad_t, pad_l
        d.ldo = pw.N
        d.act, d.alpha, d.batch = act, 1.0, 1
        d.tile, d.k_order = tile, pw.k_order
        d.out_up2 = 1 + q
        if partial is not None:
            d.gn_partial, d.gn_groups = partial.data_ptr(), gn_groups
            d.gn_nslab, d.gn_strip_off = 4 * ((IH * IW) // 64), q * ((IH * IW) // 64)
        _igemm(d, 2.0 * d.M * pw.alg_nk * 9 / 4, executed=2.0 * d.M * pw.alg_nk)
    if partial is not None:
        out._lbmk_gn_partial = (partial, gn_groups, (4 * IH * IW) // 64)
    return out


def pack_conv_weight_dgrad(weight: torch.Tensor) -> PackedWeight:
    """input-gradient filter of a stride-1 convolution: in/out channels swapped, taps flipped"""
    if weight.is_cuda and weight.dtype == bf16 and weight.is_contiguous():
        return _pack_conv_weight_dev(weight.detach(), None, True)
    return pack_conv_weight(weight.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous(), None)


def pack_linear_weight(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedWeight:
    """[N, K] -> bf16 [N, Kp] (K padded to a multiple of 64 with zeros)."""
    N, K = weight.shape
    k8 = ceil_to(K, 8)
    b = None if bias is None else f32_of(bias)
    w = weight.detach()
    if w.dtype == torch.float32:
        # fp32 mode: the f32 parameter itself when its rows are 16-byte multiples
        if K == k8 and w.is_contiguous() and w.data_ptr() % 16 == 0:
            return PackedWeight(w, b, N, k8, k8, k8, 1, 1, alg_nk=N * K)
        packed = torch.zeros(N, k8, dtype=torch.float32, device=w.device)
        packed[:, :K] = w
        return PackedWeight(packed, b, N, k8, k8, k8, 1, 1, alg_nk=N * K)
    if K == k8 and w.dtype == bf16 and w.is_contiguous() and w.data_ptr() % 16 == 0:
        # already the kernel's layout: alias the parameter (the K tail of the last 64-deep tile is masked
        # by the kernel), so an optimizer step needs no re-pack
        return PackedWeight(w, b, N, k8, k8, k8, 1, 1, alg_nk=N * K)
    Kp = ceil_to(k8, 64)
    packed = torch.zeros(N, Kp, dtype=bf16, device=weight.device)
    packed[:, :K] = w.to(bf16)
    return PackedWeight(packed, b, N, k8, Kp, k8, 1, 1, alg_nk=N * K)


def pack_geglu_weight(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedWeight:
    """GEGLU projection [2*F, K]: rows 0..F-1 are the value half, F..2F-1 the gate half
    (diffusers GEGLU: ``hidden, gate = proj(x).chunk(2, -1)``).  Rows are interleaved in
    groups of 16 (16 value rows, then their 16 gate rows) so that one lane of the GEMM
    epilogue holds a value and its gate."""
    N2, K = weight.shape
    F = N2 // 2
    if weight.dtype == torch.float32:
        # fp32 mode: plain [2F, K] projection; the gate is applied by its own kernel (lbmk_geglu_f32)
        pw = pack_linear_weight(weight, bias)
        pw.geglu = True
        return pw
    assert F % 16 == 0
    idx = torch.arange(F, device=weight.device).reshape(F // 16, 16)
    order = torch.cat([idx, idx + F], dim=1).reshape(-1)
    pw = pack_linear_weight(weight.detach()[order], None if bias is None else bias.detach()[order])
    pw.geglu = True
    return pw


# --------------------------------------------------------------------------------------
# implicit GEMM
# --------------------------------------------------------------------------------------
_profile = None   # when a dict: per-launch HIP-event timing of the implicit-GEMM kernel (bench.py)


def igemm_profile_begin():
    global _profile
    _profile = {"events": [], "flops": 0.0, "executed": 0.0}


def igemm_profile_end() -> dict:
    """Sum of the event-bracketed launch durations (ms), algorithmic FLOPs and launch count."""
    global _profile
    p, _profile = _profile, None
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in p["events"])
    return {"ms": ms, "flops": p["flops"], "executed": p["executed"], "launches": len(p["events"]), "aux": aux_profile_summary(p)}


def _timed(kind: str, work: float, launch):
    """Run ``launch()``; while a profile is open, bracket it with HIP events on the launch stream and book
    ``work`` (algorithmic bytes or FLOPs) under ``kind`` (the secondary kernels of bench.py's report)."""
    if _profile is None:
        return launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = launch()
    b.record()
    _profile.setdefault("aux", {}).setdefault(kind, []).append((a, b, work))
    return r


def aux_profile_summary(p: dict) -> dict:
    """kind -> (ms, work, launches) of the secondary kernels recorded between igemm_profile_begin / _end."""
    out = {}
    for kind, recs in p.get("aux", {}).items():
        out[kind] = (sum(a.elapsed_time(b) for a, b, _ in recs), sum(w for _, _, w in recs), len(recs))
    return out


_F32_SPLIT = os.environ.get("LBM_NO_F32_SPLIT") is None      # LBM_NO_F32_SPLIT=1: fp32 mode never splits K (A/B runs)


def _igemm(desc: IgemmDesc, alg_flops: Optional[float] = None, executed: Optional[float] = None):
    """alg_flops: products of the reference operation this launch stands for (un-padded shapes; an upsampling
    convolution counts its 9 taps); executed: products the launch multiplies, when an algebraic shortcut makes that less"""
    part = None
    if desc.in_f32 and _F32_SPLIT and lib.lbmk_igemm_f32_split_pays(desc.M, desc.N, desc.K, desc.OH * desc.OW, max(desc.batch, 1)):
        # fp32 mode, few output tiles and a deep K: one workgroup per (tile, K chunk) and a finishing pass;
        # same bits as the single launch (the library adds the chunk sums in the same order either way)
        desc.f32_chunks = lib.lbmk_igemm_f32_chunks(desc.K, desc.N, desc.OH * desc.OW)
        part = torch.empty(desc.f32_chunks * desc.M * desc.N, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
        desc.f32_part = part.data_ptr()
    if _profile is None:
        check(lib.lbmk_igemm(C.byref(desc), _stream()), "lbmk_igemm")
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()          # torch's current stream is the stream the kernel is launched on
    check(lib.lbmk_igemm(C.byref(desc), _stream()), "lbmk_igemm")
    b.record()
    _profile["events"].append((a, b))
    if alg_flops is None:
        alg_flops = 2.0 * desc.M * desc.N * desc.K * desc.batch
    _profile["flops"] += alg_flops
    _profile["executed"] += alg_flops if executed is None else executed


def conv2d(x: torch.Tensor, pw: PackedWeight, *, stride: int = 1, pad=(1, 1), up2: bool = False,
           out_hw=None, rowvec: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: int = ACT_NONE, cin_off: int = 0, tile: int = -1, gn_groups: int = 0) -> torch.Tensor:
    """x: bf16 [B, IH, IW, Cs] channels-last; reads channels [cin_off, cin_off + pw.Cin).
    pad = (top, left); the bottom/right extent follows from out_hw (default: 'same' rule
    (I + 2*pad - K)//stride + 1 with symmetric padding).
    gn_groups > 0: the epilogue also writes GroupNorm partial sums for a consumer with that many groups
    (when the shape allows); they ride on the returned tensor and `groupnorm` picks them up."""
    _need_gpu(x, pw.w)
    f32 = _act_f32(x, pw.w)
    assert x.dim() == 4 and x.is_contiguous()
    B, IH, IW, Cs = x.shape
    IHe, IWe = (2 * IH, 2 * IW) if up2 else (IH, IW)
    if out_hw is None:
        OH = (IHe + 2 * pad[0] - pw.KH) // stride + 1
        OW = (IWe + 2 * pad[1] - pw.KW) // stride + 1
    else:
        OH, OW = out_hw
    assert not (f32 and pw.geglu), "GEGLU projections are linear layers"
    n_out = pw.N // 2 if pw.geglu else pw.N
    out = torch.empty(B, OH, OW, n_out, dtype=x.dtype, device=x.device)
    d = IgemmDesc()
    d.a, d.w, d.out = x.data_ptr(), pw.w.data_ptr(), out.data_ptr()
    d.bias = _ptr(pw.bias)
    d.rowvec = _ptr(rowvec)
    d.residual = _ptr(residual)
    d.in_f32 = 1 if f32 else 0
    d.M, d.N, d.K, d.Kp = B * OH * OW, pw.N, pw.K, pw.Kp
    d.Cin, d.a_cs, d.a_coff = pw.Cin, Cs, cin_off
    d.IH, d.IW, d.OH, d.OW = IH, IW, OH, OW
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = pw.KH, pw.KW, stride, pad[0], pad[1]
    d.up2 = 1 if up2 else 0
    d.ldo = n_out
    d.ldr = residual.shape[-1] if residual is not None else 0
    d.rowvec_ld = rowvec.stride(0) if rowvec is not None else 0
    if residual is not None:
        assert residual.dtype == x.dtype and residual.is_contiguous() and residual.numel() == d.M * d.ldr
    if rowvec is not None:
        # [B, N] rows, possibly a column window of a wider matrix
        assert rowvec.dtype == x.dtype and rowvec.dim() == 2 and rowvec.stride(1) == 1
        assert rowvec.shape == (B, pw.N)
    d.act = ACT_GEGLU if pw.geglu else act
    d.out_f32, d.bias_per_m, d.alpha = (1 if f32 else 0), 0, 1.0
    d.batch, d.sa, d.sw, d.so, d.sr = 1, 0, 0, 0, 0
    d.tile = tile
    d.k_order = pw.k_order
    partial = _attach_gn_partial(d, out, gn_groups) if (gn_groups and n_out == pw.N and not f32) else None
    _igemm(d, 2.0 * d.M * pw.alg_nk if pw.alg_nk else None)
    if partial is not None:
        out._lbmk_gn_partial = partial
    return out


def _attach_gn_partial(d: IgemmDesc, out: torch.Tensor, groups: int):
    """Allocate the statistics buffer of a producer launch; None when the shape is not eligible."""
    nfl = lib.lbmk_igemm_gn_partial_floats(d.M, d.N, d.Cin, d.OH, d.OW, groups)
    if nfl <= 0 or d.ldo != d.N:
        return None
    buf = torch.empty(nfl, dtype=torch.float32, device=out.device)
    d.gn_partial, d.gn_groups = buf.data_ptr(), groups
    return (buf, groups, (d.OH * d.OW) // 64)


def linear(x: torch.Tensor, pw: PackedWeight, *, residual: Optional[torch.Tensor] = None,
           rowvec: Optional[torch.Tensor] = None, rows_per_image: int = 0, act: int = ACT_NONE,
           out: Optional[torch.Tensor] = None, tile: int = -1) -> torch.Tensor:
    """x: bf16 [..., K] with contiguous rows (row stride = x.stride(-2)); returns [..., N]."""
    _need_gpu(x, pw.w)
    f32 = _act_f32(x, pw.w)
    assert x.stride(-1) == 1
    if f32 and pw.geglu:
        # fp32 mode: projection to [.., 2F], then the gate as its own kernel
        plain = PackedWeight(pw.w, pw.bias, pw.N, pw.K, pw.Kp, pw.Cin, 1, 1, alg_nk=pw.alg_nk)
        h = linear(x, plain, tile=tile)
        assert residual is None and rowvec is None and act == ACT_NONE
        F = pw.N // 2
        y = out if out is not None else torch.empty(*x.shape[:-1], F, dtype=torch.float32, device=x.device)
        assert y.is_contiguous()
        check(lib.lbmk_geglu_f32(h.data_ptr(), y.data_ptr(), h.numel() // pw.N, F, _stream()), "lbmk_geglu_f32")
        return y
    lead = x.shape[:-1]
    M = 1
    for s in lead:
        M *= s
    Kx = x.shape[-1]
    assert Kx == pw.K and Kx % 8 == 0, (Kx, pw.K)
    lda = x.stride(-2) if x.dim() >= 2 else Kx
    # all leading dims must collapse onto one row stride
    if x.dim() > 2:
        assert x.is_contiguous() or all(x.stride(i) == x.stride(i + 1) * x.shape[i + 1] for i in range(x.dim() - 2))
    n_out = pw.N // 2 if pw.geglu else pw.N
    if out is None:
        out = torch.empty(*lead, n_out, dtype=x.dtype, device=x.device)
    assert not f32 or out.dtype == torch.float32
    d = IgemmDesc()
    d.a, d.w, d.out = x.data_ptr(), pw.w.data_ptr(), out.data_ptr()
    d.bias, d.rowvec, d.residual = _ptr(pw.bias), _ptr(rowvec), _ptr(residual)
    d.in_f32 = 1 if f32 else 0
    assert residual is None or residual.dtype == x.dtype
    d.M, d.N, d.K, d.Kp = M, pw.N, pw.K, pw.Kp
    d.Cin, d.a_cs, d.a_coff = pw.K, lda, 0
    if rowvec is not None:
        # rows of one image are consecutive: model them as a rows_per_image x 1 "image"
        assert rows_per_image > 0 and M % rows_per_image == 0
        d.IH, d.IW, d.OH, d.OW = rows_per_image, 1, rows_per_image, 1
        assert rowvec.dtype == x.dtype and rowvec.dim() == 2 and rowvec.stride(1) == 1
        d.rowvec_ld = rowvec.stride(0)
    else:
        d.IH = d.IW = d.OH = d.OW = 1
        d.rowvec_ld = 0
    d.KH = d.KW = d.stride = 1
    d.pad_t = d.pad_l = d.up2 = 0
    d.ldo = out.stride(-2) if out.dim() >= 2 else n_out
    d.ldr = residual.stride(-2) if residual is not None else 0
    d.act = ACT_GEGLU if pw.geglu else act
    d.out_f32 = 1 if out.dtype == torch.float32 else 0
    d.bias_per_m, d.alpha = 0, 1.0
    d.batch, d.sa, d.sw, d.so, d.sr = 1, 0, 0, 0, 0
    d.tile = tile
    _igemm(d, 2.0 * d.M * pw.alg_nk if pw.alg_nk else None)
    return out


_SPLITK = os.environ.get("LBM_NO_SPLITK") is None      # LBM_NO_SPLITK=1: same-box A/B runs


def splitk_chunks(M: int, N: int, K: int) -> int:
    """K chunks for a few-row linear layer (0: one launch).  Measured on MI355X: a batched launch over K chunks plus
    the finishing pass wins only when the output is too small to fill the chip (M <= 2048 rows, N <= 2560 columns)
    and K is deep (>= 3840); everywhere else it loses.  The summation order differs from the single launch, so only
    the training path (whose tile choice is not pinned to batch invariance) uses it."""
    if M > 2048 or N > 2560 or K < 3840 or not _SPLITK:
        return 0
    for S in (4, 2):
        if K % (S * 64) == 0:
            return S
    return 0


def linear_splitk(x: torch.Tensor, pw: PackedWeight, S: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: bf16 [M, K] (unit column stride) -> bf16 [M, N] = x @ W^T + bias (+ residual): S K-chunks as one batched
    implicit-GEMM launch with f32 partial outputs, then lbmk_splitk_finish."""
    _need_gpu(x, pw.w)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == bf16 and not pw.geglu and pw.KH == 1
    M, K = x.shape
    assert K == pw.K and K % (S * 64) == 0 and pw.N % 8 == 0
    Kc, N = K // S, pw.N
    part = torch.empty(S, M, N, dtype=torch.float32, device=x.device)
    d = IgemmDesc()
    d.a, d.w, d.out = x.data_ptr(), pw.w.data_ptr(), part.data_ptr()
    d.M, d.N, d.K, d.Kp = M, N, Kc, pw.Kp
    d.Cin, d.a_cs, d.a_coff = Kc, x.stride(0), 0
    d.IH = d.IW = d.OH = d.OW = 1
    d.KH = d.KW = d.stride = 1
    d.ldo, d.out_f32, d.alpha, d.act = N, 1, 1.0, ACT_NONE
    d.batch, d.sa, d.sw, d.so = S, Kc, Kc, M * N
    d.tile = -1
    _igemm(d, 2.0 * M * pw.alg_nk if pw.alg_nk else None)
    out = torch.empty(M, N, dtype=bf16, device=x.device)
    if residual is not None:
        assert residual.dtype == bf16 and residual.shape == (M, N) and residual.stride(1) == 1
    check(lib.lbmk_splitk_finish(part.data_ptr(), S, M * N, _ptr(pw.bias), _ptr(residual),
                                 residual.stride(0) if residual is not None else 0, out.data_ptr(), N, M, N, _stream()),
          "lbmk_splitk_finish")
    return out


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, out_dtype=bf16, alpha: float = 1.0,
            bias_m: Optional[torch.Tensor] = None, bias_n: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, tile: int = -1, out: Optional[torch.Tensor] = None,
            smx=None) -> torch.Tensor:
    """Batched out[z] = alpha * a[z] @ b[z]^T.  a: bf16 [M, K], [Z, M, K] or [Z1, Z2, M, K]; b likewise with N
    rows; unit stride along K, every other stride free (multiples of 8), so per-head views of a
    [B, N, H*D] tensor batch over (image, head) in one launch.  K a multiple of 8.
    bias_m: f32 [M] per output row; bias_n: f32 [N].
    ``out``: optional preallocated view of the result shape (unit column stride; other strides free)."""
    _need_gpu(a, b)
    f32 = _act_f32(a, b)
    if f32:
        out_dtype = torch.float32
    assert a.stride(-1) == 1 and b.stride(-1) == 1
    nb = max(a.dim(), b.dim()) - 2
    assert nb in (0, 1, 2) and a.dim() in (2, nb + 2) and b.dim() in (2, nb + 2)
    lead = tuple((a if a.dim() == nb + 2 else b).shape[:nb])
    M, K = a.shape[-2], a.shape[-1]
    N = b.shape[-2]
    if f32:
        assert b.shape[-1] == K and K % 4 == 0 and b.stride(-2) % 4 == 0 and a.stride(-2) % 4 == 0
    else:
        assert b.shape[-1] == K and K % 8 == 0 and b.stride(-2) % 8 == 0 and a.stride(-2) % 8 == 0
    shape = lead + (M, N)
    if smx is not None and smx[0] == 1:
        out = a          # pass 1 of the two-pass softmax stores no tile: any valid pointer
    elif out is None:
        out = torch.empty(*shape, dtype=out_dtype, device=a.device)
    else:
        assert tuple(out.shape) == shape and out.stride(-1) == 1 and out.is_cuda
        out_dtype = out.dtype
    d = IgemmDesc()
    d.a, d.w, d.out = a.data_ptr(), b.data_ptr(), out.data_ptr()
    assert not (bias_m is not None and bias_n is not None)
    d.bias = _ptr(bias_m if bias_m is not None else bias_n)
    d.bias_per_m = 1 if bias_m is not None else 0
    d.rowvec = None
    d.residual = _ptr(residual)
    d.in_f32 = 1 if f32 else 0
    d.M, d.N, d.K, d.Kp = M, N, K, b.stride(-2)
    d.Cin, d.a_cs, d.a_coff = K, a.stride(-2), 0
    d.IH = d.IW = d.OH = d.OW = 1
    d.KH = d.KW = d.stride = 1
    d.pad_t = d.pad_l = d.up2 = 0
    d.ldo, d.rowvec_ld = out.stride(-2), 0
    d.ldr = residual.stride(-2) if residual is not None else 0
    d.act, d.out_f32, d.alpha = ACT_NONE, 1 if out_dtype == torch.float32 else 0, alpha
    full_a, full_b = a.dim() == nb + 2, b.dim() == nb + 2
    if nb == 2:
        d.batch, d.batch_inner = lead[0] * lead[1], lead[1]
        d.sa, d.sa2 = (a.stride(0), a.stride(1)) if full_a else (0, 0)
        d.sw, d.sw2 = (b.stride(0), b.stride(1)) if full_b else (0, 0)
        d.so, d.so2 = out.stride(0), out.stride(1)
        assert residual is None
    else:
        d.batch = lead[0] if nb == 1 else 1
        d.sa = a.stride(0) if (nb == 1 and full_a) else 0
        d.sw = b.stride(0) if (nb == 1 and full_b) else 0
        d.so = out.stride(0) if nb == 1 else 0
    d.sr = residual.stride(0) if (residual is not None and residual.dim() == 3) else 0
    d.tile = tile
    if smx is not None:
        mode, rowmax, part = smx
        assert part.dtype == torch.float32 and part.is_contiguous() and part.numel() == max(d.batch, 1) * M * smx_parts(N)
        d.smx_mode, d.smx_part = mode, part.data_ptr()
        if mode == 1:
            d.ldo, d.so, d.so2 = N, 0, 0     # nothing is stored
        if mode == 2:
            assert rowmax.dtype == torch.float32 and rowmax.is_contiguous() and rowmax.numel() == max(d.batch, 1) * M
            d.smx_max = rowmax.data_ptr()
    if smx is not None and smx[0] == 1:
        # pass 1 of the two-pass softmax repeats the scores product of pass 2: multiplied, but not an operation of the
        # reference (softmax(q k^T) v is 4 N^2 C FLOP), so it counts as executed work only
        _igemm(d, 0.0, 2.0 * d.M * d.N * d.K * max(d.batch, 1))
    else:
        _igemm(d)
    return out


def smx_parts(N: int) -> int:
    """64-column strips per row of the two-pass softmax partials"""
    return lib.lbmk_igemm_smx_parts(N)


def attention_two_pass(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale * q k^T) v for ONE wide head (the VAE mid block: head_dim 512), q / k bf16 [B, N, C], vt bf16
    [B, C, N] -> bf16 [B, N, C].  The scores GEMM runs twice: pass 1 keeps only the row maxima, pass 2 stores
    p = exp(s - max) as bf16 plus row sums; then P V in f32 and one normalising pass.  The f32 N x N score matrix of
    the unfused path (12 bytes of traffic per score over its three kernels) is never written: 4 bytes per score."""
    B, N, C = q.shape
    assert q.dtype == bf16 and k.shape == q.shape and vt.shape == (B, C, N) and N % 8 == 0
    alpha = scale * 1.4426950408889634       # exp2 domain
    parts = smx_parts(N)
    part = torch.empty(B, N, parts, dtype=torch.float32, device=q.device)
    gemm_nt(q, k, alpha=alpha, smx=(1, None, part))
    rowmax = part.amax(dim=-1)               # strips past N hold -inf (pass 1) / 0 (pass 2)
    p = torch.empty(B, N, N, dtype=bf16, device=q.device)
    gemm_nt(q, k, alpha=alpha, out=p, smx=(2, rowmax, part))
    inv = part.sum(dim=-1).reciprocal_()
    o = gemm_nt(p, vt, out_dtype=torch.float32)
    return (o * inv.unsqueeze(-1)).to(bf16)


# --------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------
_gn_scratch = {}


def groupnorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], groups: int,
              eps: float, silu: bool) -> torch.Tensor:
    """x: bf16 (or f32: fp32 mode) [B, H, W, C] (or [B, HW, C]); gamma/beta f32 [C]."""
    _need_gpu(x)
    assert x.dtype in (bf16, torch.float32) and x.is_contiguous()
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    if x.dtype == torch.float32:
        buf = torch.empty(lib.lbmk_groupnorm_f32_scratch_floats(B, groups), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check(lib.lbmk_groupnorm_f32(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), buf.data_ptr(), B, HW, Cc, groups,
                                     eps, 1 if silu else 0, _stream()), "lbmk_groupnorm_f32")
        return y
    fused = getattr(x, "_lbmk_gn_partial", None)
    if fused is not None and fused[1] == groups and fused[0].numel() == B * groups * fused[2] * 2:
        # the producing convolution already reduced this tensor: finalize + apply only
        pbuf, _, nslab = fused
        stats = torch.empty(B * groups * 2, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        _timed("groupnorm", 4.0 * x.numel(), lambda: check(
            lib.lbmk_groupnorm_from_partials(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), pbuf.data_ptr(),
                                             nslab, stats.data_ptr(), B, HW, Cc, groups, eps, 1 if silu else 0,
                                             _stream()), "lbmk_groupnorm_from_partials"))
        return y
    nfl = lib.lbmk_groupnorm_scratch_floats(B, HW, Cc, groups)
    if nfl < 0:
        raise _C.LbmkError(f"groupnorm: unsupported shape B={B} HW={HW} C={Cc} G={groups}")
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    buf = _gn_scratch.get(key)
    if buf is None or buf.numel() < nfl:
        buf = torch.empty(max(nfl, 1 << 16), dtype=torch.float32, device=x.device)
        _gn_scratch[key] = buf
    y = torch.empty_like(x)
    _timed("groupnorm", 6.0 * x.numel(), lambda: check(
        lib.lbmk_groupnorm(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), buf.data_ptr(),
                           B, HW, Cc, groups, eps, 1 if silu else 0, _stream()), "lbmk_groupnorm"))
    return y


def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
              eps: float = 1e-5) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype in (bf16, torch.float32) and x.is_contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    if x.dtype == torch.float32:
        check(lib.lbmk_layernorm_f32(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), rows, Cc, eps, _stream()),
              "lbmk_layernorm_f32")
        return y
    _timed("layernorm", 4.0 * x.numel(), lambda: check(
        lib.lbmk_layernorm(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), rows, Cc, eps, _stream()), "lbmk_layernorm"))
    return y


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, head_dim: int,
               scale: Optional[float] = None, return_lse: bool = False):
    """q: [B, Nq, >=H*D] (last-dim view allowed), k: [B, Nk, >=H*D], vt: [B, H*D, Nk].
    ``return_lse``: also return the f32 [B, H, Nq] log2-sum-exp rows ``flash_attn_bwd`` consumes."""
    _need_gpu(q, k, vt)
    assert q.dtype == k.dtype == vt.dtype and q.dtype in (bf16, torch.float32)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and vt.stride(-1) == 1
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    if q.dtype == torch.float32:
        assert not return_lse
        return _attention_f32(q, k, vt, heads, head_dim, scale)
    o = torch.empty(B, Nq, heads * head_dim, dtype=bf16, device=q.device)
    lse = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device) if return_lse else None
    _timed("flash_attn", 4.0 * B * heads * Nq * Nk * head_dim, lambda: check(
        lib.lbmk_flash_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, heads, Nq, Nk,
                                head_dim, q.stride(1), k.stride(1), vt.stride(1), o.stride(1),
                                q.stride(0), k.stride(0), vt.stride(0), o.stride(0), scale,
                                lse.data_ptr() if return_lse else None, _stream()), "lbmk_flash_attn_fwd"))
    return (o, lse) if return_lse else o


def _attention_f32(q, k, vt, heads: int, head_dim: int, scale: float) -> torch.Tensor:
    """fp32 mode: softmax(q k^T * scale) v with the scores materialised — scores GEMM batched over (image, head),
    f32 row softmax, probabilities x V^T — in chunks of images that keep the score buffer under 2 GiB.
    Keys at or past k.shape[1] (zero columns padding V^T to a multiple of 4) get probability 0."""
    B, Nq, Nk = q.shape[0], q.shape[1], k.shape[1]
    H, D = heads, head_dim
    C = H * D
    nkp = vt.shape[2]
    assert nkp >= Nk and nkp % 4 == 0 and vt.shape[1] == C
    o = torch.empty(B, Nq, C, dtype=torch.float32, device=q.device)
    nb = max(1, min(B, (2 << 30) // (H * Nq * nkp * 4)))
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        n = b1 - b0
        qh = q[b0:b1, :, :C].unflatten(2, (H, D)).permute(0, 2, 1, 3)            # [n, H, Nq, D] views
        kh = k[b0:b1, :, :C].unflatten(2, (H, D)).permute(0, 2, 1, 3)            # [n, H, Nk, D]
        s = torch.zeros(n, H, Nq, nkp, dtype=torch.float32, device=q.device) if nkp != Nk else \
            torch.empty(n, H, Nq, nkp, dtype=torch.float32, device=q.device)
        gemm_nt(qh, kh, alpha=scale, out=s[..., :Nk] if nkp != Nk else s)
        check(lib.lbmk_softmax_rows_f32(s.data_ptr(), s.data_ptr(), n * H * Nq, Nk, nkp, nkp, _stream()), "lbmk_softmax_rows_f32")
        vh = vt[b0:b1].unflatten(1, (H, D))                                      # [n, H, D, nkp]
        gemm_nt(s, vh, out=o[b0:b1].unflatten(2, (H, D)).permute(0, 2, 1, 3))
        del s
    return o


FLASH_BWD_HEAD_DIMS = (40, 64, 80, 160)


def flash_attn_bwd_supported(head_dim: int, Nq: int, Nk: int) -> bool:
    return head_dim in FLASH_BWD_HEAD_DIMS and Nq % 64 == 0 and Nk % 64 == 0


def flash_attn_bwd(q, k, v, o, do, lse, heads: int, head_dim: int, scale: float, out=None, transposed=None):
    """Gradients of softmax(q k^T * scale) v, scores never materialised.  q / k / v: bf16 [B, N, H*D], either
    contiguous or column windows of one packed tensor (equal row and batch strides); o, do contiguous.
    ``out``: optional (dq, dk, dv) destinations with the strides of q / k / v.  ``transposed``: optional
    (qt, kt) channel-major [B, H*D, N] views when the caller already holds them.  Returns (dq, dk, dv)."""
    _need_gpu(q, k, v, o, do, lse)
    B, Nq, C = q.shape
    Nk = k.shape[1]
    assert C == heads * head_dim and flash_attn_bwd_supported(head_dim, Nq, Nk)
    for t in (q, k, v):
        assert t.dtype == bf16 and t.stride(2) == 1 and t.shape[2] == C and t.stride(1) == q.stride(1)
    assert k.stride(0) == v.stride(0) and k.shape == v.shape
    for t in (o, do):
        assert t.dtype == bf16 and t.is_contiguous() and t.shape == q.shape
    assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.shape == (B, heads, Nq)
    qt, kt = transposed if transposed is not None else (transpose(q), transpose(k))      # [B, C, N]
    dot = transpose(do)
    delta = torch.empty_like(lse)
    if out is None:
        dq, dk, dv = (torch.empty(t.shape, dtype=bf16, device=q.device) for t in (q, k, v))
        assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous(), "strided q / k / v need strided outputs"
    else:
        dq, dk, dv = out
        for g, t in ((dq, q), (dk, k), (dv, v)):
            assert g.dtype == bf16 and g.shape == t.shape and g.stride() == t.stride()
    for t in (qt, kt, dot):
        assert t.dtype == bf16 and t.stride(2) == 1 and t.shape[1] == C
    assert qt.stride(1) == dot.stride(1)
    check(lib.lbmk_flash_attn_bwd(q.data_ptr(), qt.data_ptr(), k.data_ptr(), kt.data_ptr(), v.data_ptr(),
                                  o.data_ptr(), do.data_ptr(), dot.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                  dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, heads, Nq, Nk, head_dim,
                                  q.stride(1), C, qt.stride(1), kt.stride(1), q.stride(0), k.stride(0), Nq * C,
                                  qt.stride(0), dot.stride(0), kt.stride(0), scale, _stream()), "lbmk_flash_attn_bwd")
    return dq, dk, dv


def softmax_rows(s: torch.Tensor, out_dtype=bf16) -> torch.Tensor:
    """f32 [..., cols] -> bf16 probabilities (``out_dtype=torch.float32``: fp32 mode, in place)."""
    _need_gpu(s)
    assert s.dtype == torch.float32 and s.is_contiguous()
    cols = s.shape[-1]
    rows = s.numel() // cols
    if out_dtype == torch.float32:
        check(lib.lbmk_softmax_rows_f32(s.data_ptr(), s.data_ptr(), rows, cols, cols, cols, _stream()), "lbmk_softmax_rows_f32")
        return s
    p = torch.empty(s.shape, dtype=bf16, device=s.device)
    _timed("softmax_rows", 6.0 * s.numel(), lambda: check(
        lib.lbmk_softmax_rows(s.data_ptr(), p.data_ptr(), rows, cols, cols, cols, _stream()), "lbmk_softmax_rows"))
    return p


# --------------------------------------------------------------------------------------
# bridge arithmetic
# --------------------------------------------------------------------------------------
def bridge_interp(z_src, z, eps, sigma, bridge_sigma: float, copy_src=None, out_dtype=torch.float32):
    _need_gpu(z_src, z, eps, sigma)
    assert z_src.shape == z.shape and z_src.dtype == z.dtype and z_src.is_contiguous() and z.is_contiguous()
    if eps is not None:
        assert eps.shape == z.shape and eps.dtype == z.dtype and eps.is_contiguous()
    B = z.shape[0]
    sigma = sigma.reshape(B).float().contiguous()
    if copy_src is not None:
        copy_src = copy_src.reshape(B).to(torch.uint8).contiguous()
    out = torch.empty(z.shape, dtype=out_dtype, device=z.device)
    check(lib.lbmk_bridge_interp(z_src.data_ptr(), z.data_ptr(), _ptr(eps), sigma.data_ptr(), _ptr(copy_src),
                                 out.data_ptr(), B, z[0].numel(), float(bridge_sigma), _dt(z), _dt(out), _stream()),
          "lbmk_bridge_interp")
    return out


def euler_step(x, pred, s: float, s_next: float, eps=None, bridge_sigma: float = 0.0):
    _need_gpu(x, pred, eps)
    assert x.shape == pred.shape and x.dtype == pred.dtype and x.is_contiguous() and pred.is_contiguous()
    if eps is not None:
        assert eps.shape == x.shape and eps.dtype == x.dtype and eps.is_contiguous()
    out = torch.empty_like(x)
    check(lib.lbmk_euler_step(x.data_ptr(), pred.data_ptr(), _ptr(eps), out.data_ptr(), x.numel(), float(s),
                              float(s_next), float(bridge_sigma), _dt(x), _stream()), "lbmk_euler_step")
    return out


def latent_loss(pred, z_src, z, mask=None, l1: bool = False):
    """per-sample mean of ((pred - (z_src - z)) * mask)^2 or |.|; pred may be f32 while z is bf16."""
    _need_gpu(pred, z_src, z, mask)
    assert pred.shape == z.shape == z_src.shape and z.dtype == z_src.dtype
    assert pred.is_contiguous() and z.is_contiguous() and z_src.is_contiguous()
    B = z.shape[0]
    if mask is not None:
        mask = mask.expand_as(z).to(torch.uint8).contiguous()
    out = torch.empty(B, dtype=torch.float32, device=z.device)
    check(lib.lbmk_latent_loss(pred.data_ptr(), z_src.data_ptr(), z.data_ptr(), _ptr(mask), out.data_ptr(), B,
                               z[0].numel(), 1 if l1 else 0, _dt(pred), _dt(z), _stream()), "lbmk_latent_loss")
    return out


def axpy_rows(x, y, a):
    """out[b] = x[b] + a[b] * y[b] in f32 (a: f32 [B])."""
    _need_gpu(x, y, a)
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous()
    B = x.shape[0]
    a = a.reshape(B).float().contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.lbmk_axpy_rows(x.data_ptr(), y.data_ptr(), a.data_ptr(), out.data_ptr(), B, x[0].numel(),
                             _dt(x), _dt(y), _stream()), "lbmk_axpy_rows")
    return out


# --------------------------------------------------------------------------------------
# layout / glue
# --------------------------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor, cp: Optional[int] = None, dtype=bf16) -> torch.Tensor:
    """NCHW (bf16 / f32) -> channels-last ``dtype`` (bf16, or f32 in fp32 mode), channels zero-padded to ``cp``"""
    _need_gpu(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    cp = ceil_to(Cc, 8) if cp is None else cp
    if dtype == torch.float32:
        y = torch.empty(B, H, W, cp, dtype=torch.float32, device=x.device)
        check(lib.lbmk_nchw_to_nhwc_f32(x.data_ptr(), y.data_ptr(), B, Cc, H, W, cp, _dt(x), _stream()), "lbmk_nchw_to_nhwc_f32")
        return y
    y = torch.empty(B, H, W, cp, dtype=bf16, device=x.device)
    check(lib.lbmk_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, Cc, H, W, cp, _dt(x), _stream()), "lbmk_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, channels: int, out_dtype=bf16) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype in (bf16, torch.float32) and x.is_contiguous()
    B, H, W, Cs = x.shape
    y = torch.empty(B, channels, H, W, dtype=out_dtype, device=x.device)
    if x.dtype == torch.float32:
        check(lib.lbmk_nhwc_to_nchw_f32(x.data_ptr(), y.data_ptr(), B, channels, H, W, Cs, _dt(y), _stream()),
              "lbmk_nhwc_to_nchw_f32")
        return y
    check(lib.lbmk_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), B, channels, H, W, Cs, _dt(y), _stream()),
          "lbmk_nhwc_to_nchw")
    return y


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat([a, b], channel axis) for channels-last tensors (bf16, or f32 in fp32 mode)."""
    _need_gpu(a, b)
    assert a.shape[:-1] == b.shape[:-1] and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    Ca, Cb = a.shape[-1], b.shape[-1]
    out = torch.empty(*a.shape[:-1], Ca + Cb, dtype=a.dtype, device=a.device)
    pixels = a.numel() // Ca
    if a.dtype == torch.float32:
        check(lib.lbmk_copy_channels_f32(a.data_ptr(), out.data_ptr(), pixels, Ca, Ca, 0, Ca + Cb, 0, _stream()), "copy_channels_f32")
        check(lib.lbmk_copy_channels_f32(b.data_ptr(), out.data_ptr(), pixels, Cb, Cb, 0, Ca + Cb, Ca, _stream()), "copy_channels_f32")
        return out
    check(lib.lbmk_copy_channels(a.data_ptr(), out.data_ptr(), pixels, Ca, Ca, 0, Ca + Cb, 0, _stream()), "copy_channels")
    check(lib.lbmk_copy_channels(b.data_ptr(), out.data_ptr(), pixels, Cb, Cb, 0, Ca + Cb, Ca, _stream()), "copy_channels")
    return out


def posterior_sample(moments: torch.Tensor, eps: torch.Tensor, cz: int, shift: float, scale: float) -> torch.Tensor:
    """moments: bf16 NHWC [B, H, W, ms]; eps: NCHW [B, cz, H, W] (bf16 / f32) -> z NCHW, eps.dtype."""
    _need_gpu(moments, eps)
    assert moments.dtype in (bf16, torch.float32) and moments.is_contiguous() and eps.is_contiguous()
    B, H, W, ms = moments.shape
    assert eps.shape == (B, cz, H, W)
    if moments.dtype == torch.float32:
        eps = eps.float()
        z = torch.empty_like(eps)
        check(lib.lbmk_posterior_sample_f32(moments.data_ptr(), eps.data_ptr(), z.data_ptr(), B, cz, H * W, ms, float(shift),
                                            float(scale), _stream()), "lbmk_posterior_sample_f32")
        return z
    z = torch.empty_like(eps)
    check(lib.lbmk_posterior_sample(moments.data_ptr(), eps.data_ptr(), z.data_ptr(), B, cz, H * W, ms,
                                    float(shift), float(scale), _dt(eps), _stream()), "lbmk_posterior_sample")
    return z


def timestep_embedding(t: torch.Tensor, dim: int, dtype=bf16) -> torch.Tensor:
    _need_gpu(t)
    t = t.reshape(-1).float().contiguous()
    if dtype == torch.float32:
        out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
        check(lib.lbmk_timestep_embedding_f32(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream()), "timestep_embedding_f32")
        return out
    out = torch.empty(t.shape[0], dim, dtype=bf16, device=t.device)
    check(lib.lbmk_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream()), "timestep_embedding")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype in (bf16, torch.float32) and x.is_contiguous()
    y = torch.empty_like(x)
    if x.dtype == torch.float32:
        check(lib.lbmk_silu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "lbmk_silu_f32")
        return y
    check(lib.lbmk_silu(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "lbmk_silu")
    return y


def scale_shift(x: torch.Tensor, a: float, b: float) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype in (bf16, torch.float32) and x.is_contiguous()
    y = torch.empty_like(x)
    if x.dtype == torch.float32:
        check(lib.lbmk_scale_shift_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(a), float(b), _stream()), "scale_shift_f32")
        return y
    check(lib.lbmk_scale_shift(x.data_ptr(), y.data_ptr(), x.numel(), float(a), float(b), _stream()), "scale_shift")
    return y


def tile_accumulate(tile, acc, wsum, wy, wx, th, tw, y0, x0, channels):
    _need_gpu(tile, acc, wsum, wy, wx)
    B, thf, twf, Cs = tile.shape
    _, Cc, H, W = acc.shape
    assert Cc == channels and acc.dtype == torch.float32 and wsum.shape == (H, W)
    if tile.dtype == torch.float32:
        check(lib.lbmk_tile_accumulate_f32(tile.data_ptr(), acc.data_ptr(), wsum.data_ptr(), wy.data_ptr(), wx.data_ptr(),
                                           B, Cc, Cs, thf, twf, th, tw, H, W, y0, x0, _stream()), "tile_accumulate_f32")
        return
    check(lib.lbmk_tile_accumulate(tile.data_ptr(), acc.data_ptr(), wsum.data_ptr(), wy.data_ptr(), wx.data_ptr(),
                                   B, Cc, Cs, thf, twf, th, tw, H, W, y0, x0, _stream()), "tile_accumulate")


def tile_finalize(acc, wsum, out_dtype=torch.float32):
    _need_gpu(acc, wsum)
    B, Cc, H, W = acc.shape
    out = torch.empty(B, Cc, H, W, dtype=out_dtype, device=acc.device)
    check(lib.lbmk_tile_finalize(acc.data_ptr(), wsum.data_ptr(), out.data_ptr(), B, Cc, H, W, _dt(out), _stream()),
          "tile_finalize")
    return out


# --------------------------------------------------------------------------------------
# backward (training) path: operand re-layout and reductions; the GEMMs are lbmk_igemm launches
# --------------------------------------------------------------------------------------
def transpose(x: torch.Tensor) -> torch.Tensor:
    """bf16 [R, C] or [Z, R, C] (row stride >= C) -> contiguous [C, R] / [Z, C, R]; R and C multiples of 8."""
    _need_gpu(x)
    assert x.dtype == bf16 and x.dim() in (2, 3) and x.stride(-1) == 1
    R, Cc = x.shape[-2], x.shape[-1]
    Z = x.shape[0] if x.dim() == 3 else 1
    out = torch.empty(*x.shape[:-2], Cc, R, dtype=bf16, device=x.device)
    check(lib.lbmk_transpose(x.data_ptr(), out.data_ptr(), R, Cc, x.stride(-2), R, Z,
                             x.stride(0) if x.dim() == 3 else 0, Cc * R, _stream()), "lbmk_transpose")
    return out


def colsum(x: torch.Tensor, rows_per_group: Optional[int] = None, out_dtype=torch.float32,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: bf16 [rows, N] -> [rows / rows_per_group, N] sums over each group of consecutive rows.
    ``out``: contiguous bf16 / f32 destination of that many elements (e.g. a bias's slot of a flat gradient buffer)."""
    _need_gpu(x)
    assert x.dtype == bf16 and x.dim() == 2 and x.stride(1) == 1
    rows, N = x.shape
    rpg = rows if rows_per_group is None else rows_per_group
    nfl = lib.lbmk_colsum_scratch_floats(rows, rpg, N)
    if nfl <= 0:
        raise _C.LbmkError(f"colsum: rows={rows} rows_per_group={rpg} N={N}")
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(rows // rpg, N, dtype=out_dtype, device=x.device)
    assert out.is_contiguous() and out.numel() == (rows // rpg) * N and out.dtype in (bf16, torch.float32)
    check(lib.lbmk_colsum(x.data_ptr(), out.data_ptr(), scratch.data_ptr(), rows, rpg, N, x.stride(0), _dt(out),
                          _stream()), "lbmk_colsum")
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out_dtype=torch.float32, out: Optional[torch.Tensor] = None,
            splits: Optional[int] = None) -> torch.Tensor:
    """out[n][k] = sum_m a[m][n] * b[m][k]  (weight gradient: a = dY [M, N], b = X [M, K], both row-major).
    The kernel reads both operands as they are (transposing LDS reads) and splits the row range across
    workgroups; partial products are summed in a fixed order.
    ``out``: contiguous [N, K] destination (e.g. the parameter's slot in a flat gradient buffer)."""
    _need_gpu(a, b)
    M, N = a.shape
    K = b.shape[1]
    if not (a.dtype == bf16 and b.dtype == bf16 and a.stride(1) == 1 and b.stride(1) == 1 and b.shape[0] == M
            and N % 8 == 0 and K % 8 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        return gemm_tn_t(transpose(_pad_rows8(a)), transpose(_pad_rows8(b)), out_dtype, out=out)
    if out is not None and (tuple(out.shape) != (N, K) or not out.is_contiguous() or out.dtype not in (bf16, torch.float32)):
        out.copy_(gemm_tn(a, b, torch.float32, splits=splits)[:out.shape[0], :out.shape[1]])
        return out
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    if splits is None:
        # measured: a full grid of output tiles (>= one per CU) wants no split; below that ~400 workgroups in total,
        # each with at least 512 rows of the reduction
        splits = 1 if tiles >= 256 else max(1, min(-(-400 // tiles), M // 512))
    chunk = -(-(-(-M // splits)) // 64) * 64
    splits = -(-M // chunk)                        # no empty split after rounding the chunk to whole stages
    if out is None:
        out = torch.empty(N, K, dtype=out_dtype, device=a.device)
    if splits == 1:
        check(lib.lbmk_gemm_tn(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), K, 1,
                               1 if out.dtype == torch.float32 else 0, _stream()), "lbmk_gemm_tn")
        return out
    part = torch.empty(splits, N, K, dtype=torch.float32, device=a.device)
    check(lib.lbmk_gemm_tn(a.data_ptr(), b.data_ptr(), part.data_ptr(), M, N, K, a.stride(0), b.stride(0), K, splits, 1,
                           _stream()), "lbmk_gemm_tn")
    return sum_partials(part, out=out)


def _pad_rows8(x: torch.Tensor) -> torch.Tensor:
    r = x.shape[0]
    if r % 8 == 0:
        return x
    return torch.nn.functional.pad(x, (0, 0, 0, 8 - r % 8))


_GEMM_NN = os.environ.get("LBM_NO_GEMM_NN") is None      # LBM_NO_GEMM_NN=1: input gradients through W^T copies (A/B runs)


_GEMM_NN_ALWAYS = os.environ.get("LBM_GEMM_NN_ALWAYS") is not None     # every eligible input gradient (A/B runs)


def gemm_nn_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    """operands the untransposed-weight input-gradient kernel takes: bf16, unit column stride, 16-byte aligned rows"""
    return (_GEMM_NN and a.is_cuda and a.dtype == bf16 and w.dtype == bf16 and a.dim() == 2 and w.dim() == 2
            and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[0] and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
            and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and 129 * a.stride(0) * 2 < (1 << 31))


def gemm_nn_wins(M: int, N: int, K: int) -> bool:
    """True where the untransposed-weight kernel beats 'transpose W once per step, then the implicit-GEMM kernel' for
    dX[M, K] = dY[M, N] . W[N, K].  Measured on MI355X (scripts/bench_gemm_nn.py): its 128x128 tile is bound by the
    transposing LDS reads at about 500-750 TFLOP/s, the implicit-GEMM kernel reaches 800-1240 on the same product once
    M >= 8192; at 2048 rows it is ahead only where the reduction is deep and the output narrow (N >= 3 K: it splits the
    reduction across workgroups), 84 vs 127 us at N 10240 / K 1280, 43 vs 51 us at 3840 / 1280."""
    return _GEMM_NN_ALWAYS or (M <= 4096 and N >= 3 * K)


def gemm_nn(a: torch.Tensor, w: torch.Tensor, splits: Optional[int] = None) -> torch.Tensor:
    """out[m][k] = sum_n a[m][n] * w[n][k]: the input gradient dX = dY . W of a linear layer, read from the parameter
    as it is stored ([N, K] row-major) -- no W^T copy.  a bf16 [M, N], w bf16 [N, K] -> bf16 [M, K].
    Few output tiles: the n range is split across workgroups, f32 partials summed in a fixed order."""
    _need_gpu(a, w)
    assert gemm_nn_ok(a, w)
    M, N = a.shape
    K = w.shape[1]
    tiles = ((M + 127) // 128) * ((K + 127) // 128)
    if splits is None:
        splits = 1 if tiles >= 200 else max(1, min(-(-400 // tiles), N // 512))
    chunk = -(-(-(-N // splits)) // 64) * 64
    splits = -(-N // chunk)
    if splits == 1:
        out = torch.empty(M, K, dtype=bf16, device=a.device)
        check(lib.lbmk_gemm_nn(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), w.stride(0), K, 1, 0,
                               _stream()), "lbmk_gemm_nn")
        return out
    part = torch.empty(splits, M, K, dtype=torch.float32, device=a.device)
    check(lib.lbmk_gemm_nn(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, a.stride(0), w.stride(0), K, splits, 1,
                           _stream()), "lbmk_gemm_nn")
    return sum_partials(part, bf16)


def sum_partials(part: torch.Tensor, out_dtype=torch.float32, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f32 [S, ...] -> sum over S in a fixed order, as bf16 or f32."""
    _need_gpu(part)
    assert part.dtype == torch.float32 and part.is_contiguous()
    if out is None:
        out = torch.empty(part.shape[1:], dtype=out_dtype, device=part.device)
    assert out.is_contiguous() and out.numel() == part[0].numel() and out.dtype in (bf16, torch.float32)
    check(lib.lbmk_sum_partials(part.data_ptr(), out.data_ptr(), out.numel(), part.shape[0],
                                1 if out.dtype == torch.float32 else 0, _stream()), "lbmk_sum_partials")
    return out


def gemm_tn_t(a_t: torch.Tensor, b_t: torch.Tensor, out_dtype=torch.float32, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n][k] = sum_m a_t[n][m] * b_t[k][m]; a_t bf16 [N, M], b_t bf16 [K, M] contiguous, M % 8 == 0."""
    _need_gpu(a_t, b_t)
    N, M = a_t.shape
    K = b_t.shape[0]
    assert b_t.shape[1] == M and M % 8 == 0 and a_t.is_contiguous() and b_t.is_contiguous()
    if out is not None and (tuple(out.shape) != (N, K) or not out.is_contiguous() or out.dtype not in (bf16, torch.float32)):
        out.copy_(gemm_tn_t(a_t, b_t, torch.float32)[:out.shape[0], :out.shape[1]])
        return out
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = max(1, min(256 // tiles, M // 1024))
    while S > 1 and (M % S or (M // S) % 8):
        S -= 1
    if S == 1:
        if out is not None:
            return gemm_nt(a_t, b_t, out_dtype=out.dtype, out=out)
        return gemm_nt(a_t, b_t, out_dtype=out_dtype)
    ms = M // S
    # chunk s = columns [s*ms, (s+1)*ms) of both operands: batch stride ms, row stride M
    a3 = a_t.as_strided((S, N, ms), (ms, M, 1))
    b3 = b_t.as_strided((S, K, ms), (ms, M, 1))
    part = gemm_nt(a3, b3, out_dtype=torch.float32)
    return sum_partials(part, out_dtype, out=out)


def im2col_t(x: torch.Tensor, cin: int, KH: int, KW: int, stride: int, pad, up2: bool, out_hw, cin_off: int = 0):
    """Transposed im2col of the NHWC input of a convolution: bf16 [KH*KW*cin, B*OH*OW]."""
    _need_gpu(x)
    assert x.dtype == bf16 and x.dim() == 4 and x.is_contiguous()
    B, IH, IW, Cs = x.shape
    OH, OW = out_hw
    M = B * OH * OW
    out = torch.empty(KH * KW * cin, M, dtype=bf16, device=x.device)
    check(lib.lbmk_im2col_t(x.data_ptr(), out.data_ptr(), B, IH, IW, OH, OW, KH, KW, stride, pad[0], pad[1],
                            1 if up2 else 0, cin, Cs, cin_off, M, _stream()), "lbmk_im2col_t")
    return out


def col2im(dcol: torch.Tensor, B: int, IH: int, IW: int, OH: int, OW: int, cin: int, KH: int, KW: int, stride: int,
           pad, up2: bool) -> torch.Tensor:
    """dcol: bf16 [B*OH*OW, >= KH*KW*cin] -> input gradient NHWC bf16 [B, IH, IW, cin]."""
    _need_gpu(dcol)
    assert dcol.dtype == bf16 and dcol.dim() == 2 and dcol.stride(1) == 1
    dx = torch.empty(B, IH, IW, cin, dtype=bf16, device=dcol.device)
    check(lib.lbmk_col2im(dcol.data_ptr(), dx.data_ptr(), B, IH, IW, OH, OW, KH, KW, stride, pad[0], pad[1],
                          1 if up2 else 0, cin, dcol.stride(0), cin, _stream()), "lbmk_col2im")
    return dx


def groupnorm_train(x, gamma, beta, groups: int, eps: float, silu: bool):
    """GroupNorm(+SiLU) that also returns the [B, G, 2] (mean, rstd) statistics for the backward pass."""
    _need_gpu(x)
    assert x.dtype == bf16 and x.is_contiguous()
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    nfl = lib.lbmk_groupnorm_scratch_floats(B, HW, Cc, groups)
    if nfl < 0:
        raise _C.LbmkError(f"groupnorm: unsupported shape B={B} HW={HW} C={Cc} G={groups}")
    buf = torch.empty(nfl, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check(lib.lbmk_groupnorm(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), buf.data_ptr(),
                             B, HW, Cc, groups, eps, 1 if silu else 0, _stream()), "lbmk_groupnorm")
    return y, buf[nfl - B * groups * 2:].view(B, groups, 2)


def _param_grad_outs(dg_out, db_out, Cc):
    """checks a pair of caller-supplied dgamma / dbeta destinations; -> 1 (f32) / 0 (bf16)"""
    assert dg_out is not None and db_out is not None and dg_out.dtype == db_out.dtype and dg_out.dtype in (bf16, torch.float32)
    assert dg_out.is_contiguous() and db_out.is_contiguous() and dg_out.numel() == Cc and db_out.numel() == Cc
    return 1 if dg_out.dtype == torch.float32 else 0


def groupnorm_bwd(x, dy, stats, gamma, beta, groups: int, silu: bool, add=None, dg_out=None, db_out=None):
    """-> dx bf16, dgamma f32 [C], dbeta f32 [C].  ``add``: bf16 tensor of x's shape added to dx (the gradient that
    reached x through its residual link).  ``dg_out`` / ``db_out``: write dgamma / dbeta there, in their dtype (the
    parameters' slots of a flat gradient buffer); the returned dgamma / dbeta are then None."""
    _need_gpu(x, dy, stats)
    if add is not None or dg_out is not None:
        assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous() and stats.is_contiguous()
        assert add is None or (add.dtype == bf16 and add.is_contiguous() and add.shape == x.shape)
        B, Cc = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * Cc)
        nfl = lib.lbmk_groupnorm_bwd_scratch_floats(B, HW, Cc, groups)
        if nfl < 0:
            raise _C.LbmkError(f"groupnorm_bwd: unsupported shape B={B} HW={HW} C={Cc} G={groups}")
        scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        if dg_out is None:
            dgb = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
            dg, db, pf, ret = dgb[0], dgb[1], 1, (dgb[0], dgb[1])
        else:
            dg, db, pf, ret = dg_out, db_out, _param_grad_outs(dg_out, db_out, Cc), (None, None)
        check(lib.lbmk_groupnorm_bwd_ex(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), stats.data_ptr(), _ptr(gamma), _ptr(beta),
                                        dg.data_ptr(), db.data_ptr(), pf, _ptr(add), scratch.data_ptr(), B, HW, Cc, groups,
                                        1 if silu else 0, _stream()), "lbmk_groupnorm_bwd_ex")
        return (dx, *ret)
    assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous() and stats.is_contiguous()
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    nfl = lib.lbmk_groupnorm_bwd_scratch_floats(B, HW, Cc, groups)
    if nfl < 0:
        raise _C.LbmkError(f"groupnorm_bwd: unsupported shape B={B} HW={HW} C={Cc} G={groups}")
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgb = torch.empty(2, Cc, dtype=torch.float32, device=x.device)     # rows of one tensor: one cast converts both
    dg, db = dgb[0], dgb[1]
    check(lib.lbmk_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), stats.data_ptr(), _ptr(gamma), _ptr(beta),
                                 dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), B, HW, Cc, groups, 1 if silu else 0,
                                 _stream()), "lbmk_groupnorm_bwd")
    return dx, dg, db


def layernorm_bwd(x, dy, gamma, eps: float = 1e-5, add=None, dg_out=None, db_out=None):
    """-> dx bf16, dgamma f32 [C], dbeta f32 [C]; ``add`` / ``dg_out`` / ``db_out`` as in groupnorm_bwd"""
    _need_gpu(x, dy)
    if add is not None or dg_out is not None:
        assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous()
        assert add is None or (add.dtype == bf16 and add.is_contiguous() and add.shape == x.shape)
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        nfl = lib.lbmk_layernorm_bwd_scratch_floats(rows, Cc)
        if nfl < 0:
            raise _C.LbmkError(f"layernorm_bwd: unsupported shape rows={rows} C={Cc}")
        scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        if dg_out is None:
            dgb = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
            dg, db, pf, ret = dgb[0], dgb[1], 1, (dgb[0], dgb[1])
        else:
            dg, db, pf, ret = dg_out, db_out, _param_grad_outs(dg_out, db_out, Cc), (None, None)
        check(lib.lbmk_layernorm_bwd_ex(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _ptr(gamma), dg.data_ptr(), db.data_ptr(), pf,
                                        _ptr(add), scratch.data_ptr(), rows, Cc, eps, _stream()), "lbmk_layernorm_bwd_ex")
        return (dx, *ret)
    assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    nfl = lib.lbmk_layernorm_bwd_scratch_floats(rows, Cc)
    if nfl < 0:
        raise _C.LbmkError(f"layernorm_bwd: unsupported shape rows={rows} C={Cc}")
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgb = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
    check(lib.lbmk_layernorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _ptr(gamma), dgb.data_ptr(),
                                 scratch.data_ptr(), rows, Cc, eps, _stream()), "lbmk_layernorm_bwd")
    return dx, dgb[0], dgb[1]


def silu_bwd(x, dy):
    _need_gpu(x, dy)
    assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous() and x.numel() % 8 == 0
    dx = torch.empty_like(x)
    check(lib.lbmk_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "lbmk_silu_bwd")
    return dx


def geglu_fwd(h):
    _need_gpu(h)
    assert h.dtype == bf16 and h.is_contiguous() and h.shape[-1] % 16 == 0
    F = h.shape[-1] // 2
    y = torch.empty(*h.shape[:-1], F, dtype=bf16, device=h.device)
    check(lib.lbmk_geglu_fwd(h.data_ptr(), y.data_ptr(), h.numel() // (2 * F), F, _stream()), "lbmk_geglu_fwd")
    return y


def geglu_bwd(h, dy):
    _need_gpu(h, dy)
    assert h.dtype == bf16 and dy.dtype == bf16 and h.is_contiguous() and dy.is_contiguous()
    F = h.shape[-1] // 2
    dh = torch.empty_like(h)
    check(lib.lbmk_geglu_bwd(h.data_ptr(), dy.data_ptr(), dh.data_ptr(), h.numel() // (2 * F), F, _stream()), "lbmk_geglu_bwd")
    return dh


def softmax_bwd_rows(p, dp, scale: float):
    """p bf16 [..., cols], dp f32 same shape -> dS bf16 = scale * p * (dp - rowsum(dp * p))"""
    _need_gpu(p, dp)
    assert p.dtype == bf16 and dp.dtype == torch.float32 and p.is_contiguous() and dp.is_contiguous() and p.shape == dp.shape
    cols = p.shape[-1]
    ds = torch.empty_like(p)
    check(lib.lbmk_softmax_bwd_rows(p.data_ptr(), dp.data_ptr(), ds.data_ptr(), p.numel() // cols, cols, scale, _stream()),
          "lbmk_softmax_bwd_rows")
    return ds


def latent_loss_bwd(pred, z_src, z, mask, gout, l1: bool):
    _need_gpu(pred, z_src, z, gout)
    B = pred.shape[0]
    per = pred.numel() // B
    pred, z_src, z = pred.contiguous(), z_src.contiguous(), z.contiguous()
    m = None if mask is None else mask.to(torch.uint8).expand_as(pred).contiguous()
    gout = gout.reshape(B).float().contiguous()
    dpred = torch.empty_like(pred)
    check(lib.lbmk_latent_loss_bwd(pred.data_ptr(), z_src.data_ptr(), z.data_ptr(), _ptr(m), gout.data_ptr(), dpred.data_ptr(),
                                   B, per, 1 if l1 else 0, _dt(pred), _dt(z), _stream()), "lbmk_latent_loss_bwd")
    return dpred


def adamw_step(p_flat, g_flat, m, v, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
               step: int = 1, grad_scale: float = 1.0):
    """In-place AdamW update of a flat bf16 parameter range (f32 moments)."""
    _need_gpu(p_flat, g_flat, m, v)
    assert p_flat.dtype == bf16 and g_flat.dtype == bf16 and m.dtype == torch.float32 and v.dtype == torch.float32
    n = p_flat.numel()
    assert g_flat.numel() == n and m.numel() == n and v.numel() == n
    assert p_flat.is_contiguous() and g_flat.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    check(lib.lbmk_adamw(p_flat.data_ptr(), g_flat.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, betas[0], betas[1], eps,
                         weight_decay, step, grad_scale, _stream()), "lbmk_adamw")


def _mask_u8(mask, like):
    return None if mask is None else mask.to(torch.uint8).expand_as(like).contiguous()


def pixel_loss(dec, target, mask=None, l1: bool = False, clamp: bool = True):
    """per-sample mean of ((c(dec) - target) * mask)^2 or |.| -> f32 [B]; c clips to [-1, 1] when ``clamp``"""
    _need_gpu(dec, target, mask)
    assert dec.shape == target.shape and dec.is_contiguous() and target.is_contiguous()
    B = dec.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=dec.device)
    check(lib.lbmk_pixel_loss(dec.data_ptr(), target.data_ptr(), _ptr(_mask_u8(mask, dec)), out.data_ptr(), B,
                              dec[0].numel(), 1 if l1 else 0, 1 if clamp else 0, _dt(dec), _dt(target), _stream()),
          "lbmk_pixel_loss")
    return out


def pixel_loss_bwd(dec, target, mask, gout, l1: bool, clamp: bool = True):
    _need_gpu(dec, target, gout)
    B = dec.shape[0]
    gout = gout.reshape(B).float().contiguous()
    ddec = torch.empty_like(dec)
    check(lib.lbmk_pixel_loss_bwd(dec.data_ptr(), target.data_ptr(), _ptr(_mask_u8(mask, dec)), gout.data_ptr(),
                                  ddec.data_ptr(), B, dec[0].numel(), 1 if l1 else 0, 1 if clamp else 0, _dt(dec),
                                  _dt(target), _stream()), "lbmk_pixel_loss_bwd")
    return ddec


def scale_rows(y, a):
    """out[b] = a[b] * y[b] in f32 (a: f32 [B])"""
    return axpy_rows(torch.zeros_like(y), y, a)


# --------------------------------------------------------------------------------------
# LPIPS pieces (the VGG convolutions are conv2d(..., act=ACT_RELU))
# --------------------------------------------------------------------------------------
def _f32x3(v):
    return (C.c_float * 3)(*[float(a) for a in v])


def _mask_f32(mask, B, H, W):
    if mask is None:
        return None
    m = mask.to(torch.float32).expand(B, 1, H, W).contiguous()
    return m


def lpips_prep(x: torch.Tensor, mask, shift, scale, clamp: bool) -> torch.Tensor:
    """NCHW [B, 3, H, W] (bf16 / f32) -> NHWC bf16 [B, H, W, 8]: ((clip(x) if clamp else x) * mask - shift) / scale."""
    _need_gpu(x)
    assert x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous() and x.dtype in (bf16, torch.float32)
    B, _, H, W = x.shape
    m = _mask_f32(mask, B, H, W)
    y = torch.empty(B, H, W, 8, dtype=bf16, device=x.device)
    check(lib.lbmk_lpips_prep(x.data_ptr(), _ptr(m), y.data_ptr(), B, H, W, 1 if x.dtype == torch.float32 else 0,
                              1 if clamp else 0, _f32x3(shift), _f32x3([1.0 / s for s in scale]), _stream()), "lbmk_lpips_prep")
    return y


def lpips_prep_bwd(x, mask, dy, scale, clamp: bool) -> torch.Tensor:
    _need_gpu(x, dy)
    B, _, H, W = x.shape
    assert dy.dtype == bf16 and dy.is_contiguous() and tuple(dy.shape) == (B, H, W, 8)
    m = _mask_f32(mask, B, H, W)
    dx = torch.empty_like(x)
    check(lib.lbmk_lpips_prep_bwd(x.data_ptr(), _ptr(m), dy.data_ptr(), dx.data_ptr(), B, H, W,
                                  1 if x.dtype == torch.float32 else 0, 1 if clamp else 0,
                                  _f32x3([1.0 / s for s in scale]), _stream()), "lbmk_lpips_prep_bwd")
    return dx


def relu_bwd(y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _need_gpu(y, dy)
    assert y.dtype == bf16 and dy.dtype == bf16 and y.is_contiguous() and dy.is_contiguous() and y.shape == dy.shape
    dx = torch.empty_like(y)
    check(lib.lbmk_relu_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), _stream()), "lbmk_relu_bwd")
    return dx


def maxpool2(x: torch.Tensor) -> torch.Tensor:
    """NHWC bf16 [B, H, W, C] -> [B, H/2, W/2, C]"""
    _need_gpu(x)
    assert x.dtype == bf16 and x.dim() == 4 and x.is_contiguous()
    B, H, W, Cc = x.shape
    y = torch.empty(B, H // 2, W // 2, Cc, dtype=bf16, device=x.device)
    check(lib.lbmk_maxpool2(x.data_ptr(), y.data_ptr(), B, H, W, Cc, _stream()), "lbmk_maxpool2")
    return y


def maxpool2_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _need_gpu(x, dy)
    B, H, W, Cc = x.shape
    assert dy.dtype == bf16 and dy.is_contiguous() and tuple(dy.shape) == (B, H // 2, W // 2, Cc)
    dx = torch.empty_like(x)
    check(lib.lbmk_maxpool2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, Cc, _stream()), "lbmk_maxpool2_bwd")
    return dx


def lpips_head(f0: torch.Tensor, f1: torch.Tensor, w: torch.Tensor, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f0, f1: NHWC bf16 feature maps, w: f32 [C] -> f32 [B]: spatial mean of sum_c w_c (n0_c - n1_c)^2."""
    _need_gpu(f0, f1, w)
    assert f0.dtype == bf16 and f1.dtype == bf16 and f0.is_contiguous() and f1.is_contiguous() and f0.shape == f1.shape
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == f0.shape[-1]
    B, Cc = f0.shape[0], f0.shape[-1]
    HW = f0.numel() // (B * Cc)
    assert keep is None or (keep.dtype == torch.uint8 and keep.is_contiguous() and keep.numel() == f0.numel())
    nfl = lib.lbmk_lpips_head_partial_floats(B, HW, Cc)
    if nfl <= 0:
        raise _C.LbmkError(f"lpips_head: B={B} HW={HW} C={Cc}")
    part = torch.empty(nfl // B, B, dtype=torch.float32, device=f0.device)
    check(lib.lbmk_lpips_head(f0.data_ptr(), f1.data_ptr(), w.data_ptr(), _ptr(keep), part.data_ptr(), B, HW, Cc, _stream()),
          "lbmk_lpips_head")
    return sum_partials(part) * (1.0 / HW)


def lpips_head_bwd(f0, f1, w, keep, gout: torch.Tensor) -> torch.Tensor:
    _need_gpu(f0, f1, w, gout)
    B, Cc = f0.shape[0], f0.shape[-1]
    HW = f0.numel() // (B * Cc)
    g = gout.to(torch.float32).reshape(B).contiguous()
    df0 = torch.empty_like(f0)
    check(lib.lbmk_lpips_head_bwd(f0.data_ptr(), f1.data_ptr(), w.data_ptr(), _ptr(keep), g.data_ptr(), df0.data_ptr(),
                                  B, HW, Cc, _stream()), "lbmk_lpips_head_bwd")
    return df0
