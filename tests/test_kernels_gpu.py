"""Kernel-level parity: every HIP kernel, called through the C-ABI, against a plain
PyTorch fp32 computation of the same op on the same (bf16-rounded) inputs.

Tolerances (relative L2 unless noted): bf16-output kernels 4e-3 (one bf16 rounding of the
result is 2^-9 relative worst case, ~1.7e-3 rms); f32-output kernels 1e-5.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def randbf(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(BF)


@pytest.fixture(scope="module")
def ops(cuda):
    from lbm_amd import ops as _ops
    return _ops


@pytest.fixture()
def gen():
    return torch.Generator().manual_seed(1234)


def _conv_ref(x_nhwc, w, b, stride, padding, up2=False):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if up2:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w.float(), None if b is None else b.float(), stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tile", [-1, 0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19])
@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (128, 128, 64), (17, 320, 1280), (1000, 72, 320)])
def test_linear(ops, cuda, gen, tile, M, N, K):
    x = randbf(gen, M, K)
    w = randbf(gen, N, K, scale=K ** -0.5)
    b = randbf(gen, N)
    pw = ops.pack_linear_weight(w.to(cuda), b.to(cuda))
    y = ops.linear(x.to(cuda), pw, tile=tile)
    ref = x.float() @ w.float().t() + b.float()
    assert y.shape == (M, N)
    assert rel_l2(y, ref) < 4e-3


def test_bias_load_forms_and_image_index_forms_agree(ops, cuda, gen):
    """The epilogue reads the bias with 16-byte loads when its pointer allows and element by element otherwise, and
    decomposes pixel indices with shifts for power-of-two image sizes and divisions otherwise: same results."""
    import dataclasses
    M, N, K = 600, 320, 256
    x = randbf(gen, M, K).to(cuda)
    w = randbf(gen, N, K, scale=K ** -0.5)
    b = randbf(gen, N)
    pw = ops.pack_linear_weight(w.to(cuda), b.to(cuda))
    big = torch.zeros(N + 8, dtype=torch.float32, device=cuda)
    big[1:N + 1] = pw.bias
    pw_odd = dataclasses.replace(pw, bias=big[1:N + 1])        # 4-byte aligned only
    assert pw.bias.data_ptr() % 16 == 0 and pw_odd.bias.data_ptr() % 16 == 4
    for tile in (-1, 0, 7, 11, 17, 18):
        assert torch.equal(ops.linear(x, pw, tile=tile), ops.linear(x, pw_odd, tile=tile)), tile
    # 3x3 convolution with a per-image row vector on a 24 x 40 image (divisions) against the fp32 reference;
    # the power-of-two sizes of the other tests take the shift form
    B, H, W, C, Nc = 2, 24, 40, 64, 64
    xi = randbf(gen, B, H, W, C)
    wc = randbf(gen, Nc, C, 3, 3, scale=(9 * C) ** -0.5)
    bc = randbf(gen, Nc)
    rv = randbf(gen, B, Nc)
    pc = ops.pack_conv_weight(wc.to(cuda), bc.to(cuda))
    ref = F.conv2d(xi.float().permute(0, 3, 1, 2), wc.float(), bc.float(), padding=1) + rv.float()[:, :, None, None]
    for tile in (-1, 0, 7, 15, 17):
        y = ops.conv2d(xi.to(cuda), pc, rowvec=rv.to(cuda), tile=tile)
        assert rel_l2(y.float().permute(0, 3, 1, 2), ref) < 4e-3, tile


def test_linear_small_n_tile(ops, cuda, gen):
    x = randbf(gen, 700, 320)
    w = randbf(gen, 4, 320, scale=320 ** -0.5)
    pw = ops.pack_linear_weight(w.to(cuda), None)
    y = ops.linear(x.to(cuda), pw)          # auto -> 256x16 tile
    assert rel_l2(y, x.float() @ w.float().t()) < 4e-3


def test_linear_epilogues(ops, cuda, gen):
    B, R, K, N = 3, 50, 128, 192
    x = randbf(gen, B * R, K)
    w = randbf(gen, N, K, scale=K ** -0.5)
    b = randbf(gen, N)
    res = randbf(gen, B * R, N)
    rv = randbf(gen, B, N)
    pw = ops.pack_linear_weight(w.to(cuda), b.to(cuda))
    y = ops.linear(x.to(cuda), pw, residual=res.to(cuda), rowvec=rv.to(cuda), rows_per_image=R, act=ops.ACT_SILU)
    ref = x.float() @ w.float().t() + b.float() + rv.float().repeat_interleave(R, 0) + res.float()
    ref = F.silu(ref)
    assert rel_l2(y, ref) < 4e-3
    # f32 output
    out = torch.empty(B * R, N, dtype=torch.float32, device=cuda)
    ops.linear(x.to(cuda), pw, out=out)
    assert rel_l2(out, x.float() @ w.float().t() + b.float()) < 1e-5


def test_geglu(ops, cuda, gen):
    M, K, Fh = 200, 320, 1280
    x = randbf(gen, M, K)
    w = randbf(gen, 2 * Fh, K, scale=K ** -0.5)
    b = randbf(gen, 2 * Fh)
    pw = ops.pack_geglu_weight(w.to(cuda), b.to(cuda))
    for tile in (-1, 0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 16, 17, 19):
        y = ops.linear(x.to(cuda), pw, tile=tile)
        h = x.float() @ w.float().t() + b.float()
        ref = h[:, :Fh] * F.gelu(h[:, Fh:])
        assert y.shape == (M, Fh)
        assert rel_l2(y, ref) < 4e-3, tile


@pytest.mark.parametrize("cin,cout,H,W,stride", [(64, 96, 20, 20, 1), (128, 128, 17, 23, 1), (320, 320, 16, 16, 2),
                                                 (192, 64, 9, 11, 2), (8, 128, 33, 31, 1), (72, 64, 14, 10, 1),
                                                 (96, 96, 11, 13, 2)])
def test_conv3x3(ops, cuda, gen, cin, cout, H, W, stride):
    B = 2
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, stride=stride, pad=(1, 1))
    ref = _conv_ref(x, w, b, stride, 1)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 4e-3


def test_conv_fused_epilogue(ops, cuda, gen):
    B, H, W, cin, cout = 2, 12, 12, 128, 256
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    rv = randbf(gen, B, cout)
    res = randbf(gen, B, H, W, cout)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, rowvec=rv.to(cuda), residual=res.to(cuda))
    ref = _conv_ref(x, w, b, 1, 1) + rv.float()[:, None, None, :] + res.float()
    assert rel_l2(y, ref) < 4e-3


def test_conv_asym_downsample(ops, cuda, gen):
    # VAE encoder downsample: F.pad(x, (0,1,0,1)) then 3x3 stride 2 padding 0
    B, H, W, c = 2, 16, 24, 128
    x = randbf(gen, B, H, W, c)
    w = randbf(gen, c, c, 3, 3, scale=(9 * c) ** -0.5)
    b = randbf(gen, c)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, stride=2, pad=(0, 0), out_hw=(H // 2, W // 2))
    xr = F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1))
    ref = F.conv2d(xr, w.float(), b.float(), stride=2).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 4e-3


@pytest.mark.parametrize("c", [64, 40, 192])
def test_conv_upsample_fused(ops, cuda, gen, c):
    B, H, W = 2, 9, 7
    x = randbf(gen, B, H, W, c)
    w = randbf(gen, c, c, 3, 3, scale=(9 * c) ** -0.5)
    b = randbf(gen, c)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, up2=True)
    ref = _conv_ref(x, w, b, 1, 1, up2=True)
    assert y.shape == ref.shape == (B, 2 * H, 2 * W, c)
    assert rel_l2(y, ref) < 4e-3


@pytest.mark.parametrize("cin,cout", [(4, 320), (3, 128), (12, 320), (320, 4), (128, 3), (512, 8)])
def test_conv_thin_channels(ops, cuda, gen, cin, cout):
    B, H, W = 2, 16, 16
    x_nchw = (torch.randn(B, cin, H, W, generator=gen)).to(BF)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    x = ops.nchw_to_nhwc(x_nchw.to(cuda))
    assert x.shape[-1] % 8 == 0
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x, pw)
    ref = F.conv2d(x_nchw.float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert y.shape[-1] == (cout + 7) // 8 * 8          # output channels are padded to 8 with zeros
    assert float(y[..., cout:].float().abs().sum()) == 0.0
    assert rel_l2(y[..., :cout], ref) < 4e-3
    back = ops.nhwc_to_nchw(y, cout, out_dtype=torch.float32)
    assert rel_l2(back, ref.permute(0, 3, 1, 2)) < 4e-3


def test_conv1x1_and_channel_window(ops, cuda, gen):
    B, H, W, cs, cin, cout = 2, 8, 8, 256, 128, 64
    x = randbf(gen, B, H, W, cs)
    w = randbf(gen, cout, cin, 1, 1, scale=cin ** -0.5)
    pw = ops.pack_conv_weight(w.to(cuda), None)
    y = ops.conv2d(x.to(cuda), pw, pad=(0, 0), cin_off=64)
    ref = F.conv2d(x.float()[..., 64:64 + cin].permute(0, 3, 1, 2), w.float()).permute(0, 2, 3, 1)
    assert rel_l2(y, ref) < 4e-3


def test_gemm_nt_batched(ops, cuda, gen):
    Z, M, N, K = 3, 130, 70, 128
    a = randbf(gen, Z, M, K)
    b = randbf(gen, Z, N, K)
    bm = torch.randn(M, generator=gen)
    y = ops.gemm_nt(a.to(cuda), b.to(cuda), out_dtype=torch.float32, alpha=0.125, bias_m=bm.to(cuda))
    ref = 0.125 * torch.einsum("zmk,znk->zmn", a.float(), b.float()) + bm[None, :, None]
    assert rel_l2(y, ref) < 1e-5
    # shared left operand (weights), per-batch right operand: V^T = Wv X^T
    wv = randbf(gen, M, K)
    y2 = ops.gemm_nt(wv.to(cuda), b.to(cuda))
    ref2 = torch.einsum("mk,znk->zmn", wv.float(), b.float())
    assert rel_l2(y2, ref2) < 4e-3


@pytest.mark.parametrize("tile", [7, 17, 18, 19])
def test_gemm_nt_batched_on_every_wide_tile(ops, cuda, gen, tile):
    """batched launches (grid.z, two batch levels) on the 16-wave and the ping-pong tiles: ragged M / N, several tiles
    per problem, f32 and bf16 outputs; the same bits whichever tile runs"""
    Z1, Z2, M, N, K = 2, 3, 300, 410, 192
    a = randbf(gen, Z1, Z2, M, K).to(cuda)
    b = randbf(gen, Z1, Z2, N, K).to(cuda)
    y = ops.gemm_nt(a, b, out_dtype=torch.float32, alpha=0.25, tile=tile)
    ref = 0.25 * torch.einsum("uzmk,uznk->uzmn", a.float().cpu(), b.float().cpu())
    assert rel_l2(y, ref) < 1e-5
    yb = ops.gemm_nt(a, b, tile=tile)
    assert rel_l2(yb, 4.0 * ref) < 4e-3
    assert torch.equal(yb, ops.gemm_nt(a, b, tile=0))


@pytest.mark.parametrize("C,HW", [(320, 256), (1920, 64), (2560, 36), (128, 1000), (512, 144), (960, 100)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(ops, cuda, gen, C, HW, silu):
    B = 3
    x = (torch.randn(B, HW, C, generator=gen) * 2 + 0.5).to(BF)
    g = torch.randn(C, generator=gen)
    bt = torch.randn(C, generator=gen)
    y = ops.groupnorm(x.to(cuda), g.to(cuda), bt.to(cuda), 32, 1e-5, silu)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, g, bt, 1e-5)
    if silu:
        ref = F.silu(ref)
    assert rel_l2(y, ref.permute(0, 2, 1)) < 4e-3


@pytest.mark.parametrize("C", [320, 640, 1280, 2048])
def test_layernorm(ops, cuda, gen, C):
    x = (torch.randn(37, C, generator=gen) * 3 - 1).to(BF)
    g = torch.randn(C, generator=gen)
    bt = torch.randn(C, generator=gen)
    y = ops.layernorm(x.to(cuda), g.to(cuda), bt.to(cuda), 1e-5)
    assert rel_l2(y, F.layer_norm(x.float(), (C,), g, bt, 1e-5)) < 4e-3


@pytest.mark.parametrize("Nq,Nk", [(256, 256), (200, 200), (1024, 1024), (72, 136)])
def test_flash_attn_d64(ops, cuda, gen, Nq, Nk):
    B, H, D = 2, 3, 64
    q = randbf(gen, B, Nq, H * D)
    k = randbf(gen, B, Nk, H * D)
    v = randbf(gen, B, Nk, H * D)
    vt = v.permute(0, 2, 1).contiguous()
    o = ops.flash_attn(q.to(cuda), k.to(cuda), vt.to(cuda), H, D)
    qh = q.float().view(B, Nq, H, D).transpose(1, 2)
    kh = k.float().view(B, Nk, H, D).transpose(1, 2)
    vh = v.float().view(B, Nk, H, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Nq, H * D)
    assert rel_l2(o, ref) < 6e-3


@pytest.mark.parametrize("D", [40, 80, 160])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (200, 136), (64, 5)])
def test_flash_attn_other_head_dims(ops, cuda, gen, D, Nq, Nk):
    """head dims of the SD1.5-style UNet (8 heads over 320/640/1280 channels) and the 5-token
    lighting cross-attention (keys past Nk masked, V^T padded to 8 keys)."""
    B, H = 2, 2
    q = randbf(gen, B, Nq, H * D)
    k = randbf(gen, B, Nk, H * D)
    v = randbf(gen, B, Nk, H * D)
    nkp = (Nk + 7) // 8 * 8
    vt = torch.zeros(B, H * D, nkp, dtype=BF)
    vt[:, :, :Nk] = v.permute(0, 2, 1)
    o = ops.flash_attn(q.to(cuda), k.to(cuda), vt.to(cuda), H, D)
    qh = q.float().view(B, Nq, H, D).transpose(1, 2)
    kh = k.float().view(B, Nk, H, D).transpose(1, 2)
    vh = v.float().view(B, Nk, H, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Nq, H * D)
    assert rel_l2(o, ref) < 6e-3


def test_flash_attn_online_rescale(ops, cuda, gen):
    # a spiked key late in the sequence forces the running max to jump after many tiles
    B, H, D, N = 1, 1, 64, 512
    q = randbf(gen, B, N, D)
    k = randbf(gen, B, N, D)
    v = randbf(gen, B, N, D)
    k[0, 400] = q[0, 7] * 4.0
    k[0, 130] = q[0, 300] * 3.0
    vt = v.permute(0, 2, 1).contiguous()
    o = ops.flash_attn(q.to(cuda), k.to(cuda), vt.to(cuda), H, D)
    ref = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    assert float((o.float().cpu() - ref).abs().max()) < 3e-2
    assert rel_l2(o, ref) < 6e-3


def test_flash_attn_strided_views(ops, cuda, gen):
    # q and k as column windows of one fused [B, N, 2C] projection output
    B, H, D, N = 2, 2, 64, 192
    qk = randbf(gen, B, N, 2 * H * D).to(cuda)
    v = randbf(gen, B, N, H * D)
    vt = v.permute(0, 2, 1).contiguous().to(cuda)
    o = ops.flash_attn(qk[..., :H * D], qk[..., H * D:], vt, H, D)
    qh = qk[..., :H * D].float().cpu().view(B, N, H, D).transpose(1, 2)
    kh = qk[..., H * D:].float().cpu().view(B, N, H, D).transpose(1, 2)
    vh = v.float().view(B, N, H, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * D)
    assert rel_l2(o, ref) < 6e-3


@pytest.mark.parametrize("rows,cols", [(50, 1024), (7, 1000), (5, 4096), (3, 16384), (3, 20000), (2, 65536), (2, 70000)])
def test_softmax_rows(ops, cuda, gen, rows, cols):
    """every row-length class: register-resident rows (256 threads up to 16k columns, 1024 threads up to 64k) and the
    three-pass kernel beyond"""
    s = torch.randn(rows, cols, generator=gen) * 5
    p = ops.softmax_rows(s.to(cuda))
    assert rel_l2(p, torch.softmax(s, -1)) < 4e-3
    assert (p.float().sum(-1).cpu() - 1).abs().max() < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_bridge_kernels(ops, cuda, gen, dtype):
    B, shape = 4, (4, 4, 16, 16)
    zs = torch.randn(*shape, generator=gen).to(dtype)
    z = torch.randn(*shape, generator=gen).to(dtype)
    eps = torch.randn(*shape, generator=gen).to(dtype)
    sig = torch.tensor([0.25, 0.5, 0.75, 1.0])
    bs = 0.005
    out = ops.bridge_interp(zs.to(cuda), z.to(cuda), eps.to(cuda), sig.to(cuda), bs,
                            copy_src=torch.tensor([0, 0, 0, 1]).to(cuda))
    s4 = sig.view(B, 1, 1, 1)
    ref = s4 * zs + (1.0 - s4) * z + bs * (s4 * (1.0 - s4)) ** 0.5 * eps
    ref[3] = zs[3]
    assert out.dtype == torch.float32
    assert rel_l2(out, ref) < 1e-6
    # Euler step + re-noise
    x = zs
    pred = z
    o2 = ops.euler_step(x.to(cuda), pred.to(cuda), 1.0, 0.75, eps=eps.to(cuda), bridge_sigma=bs)
    stepped = (x.float() + (0.75 - 1.0) * pred.float()).to(dtype)
    ref2 = (stepped + bs * (0.75 * 0.25) ** 0.5 * eps.float()).to(dtype)
    assert o2.dtype == dtype
    assert rel_l2(o2, ref2) < (1e-6 if dtype == torch.float32 else 3e-3)
    # latent loss
    mask = (torch.rand(*shape, generator=gen) > 0.3)
    for l1 in (False, True):
        lo = ops.latent_loss(pred.to(cuda), zs.to(cuda), z.to(cuda), mask.to(cuda), l1=l1)
        tgt = (zs - z).float()
        d = pred.float() * mask - tgt * mask
        refl = (d.abs() if l1 else d ** 2).reshape(B, -1).mean(1)
        assert rel_l2(lo, refl) < 1e-5


def test_posterior_and_misc(ops, cuda, gen):
    B, H, W = 2, 8, 8
    mom = randbf(gen, B, H, W, 8)
    eps = torch.randn(B, 4, H, W, generator=gen)
    z = ops.posterior_sample(mom.to(cuda), eps.to(cuda), 4, 0.0, 0.13025)
    m = mom.float().permute(0, 3, 1, 2)
    ref = (m[:, :4] + torch.exp(0.5 * m[:, 4:].clamp(-30, 20)) * eps) * 0.13025
    assert rel_l2(z, ref) < 1e-5
    t = torch.tensor([1000.0, 250.0, 0.5])
    emb = ops.timestep_embedding(t.to(cuda), 320)
    half = 160
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half).float() / half)
    arg = t[:, None] * freqs[None]
    refe = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    assert float((emb.float().cpu() - refe).abs().max()) < 1e-2
    x = randbf(gen, 4, 64)
    assert rel_l2(ops.silu(x.to(cuda)), F.silu(x.float())) < 4e-3
    assert rel_l2(ops.scale_shift(x.to(cuda), 1 / 0.13025, 0.25), x.float() / 0.13025 + 0.25) < 4e-3
    a = randbf(gen, 2, 5, 5, 64)
    b = randbf(gen, 2, 5, 5, 128)
    assert torch.equal(ops.concat_channels(a.to(cuda), b.to(cuda)).cpu(), torch.cat([a, b], -1))


def test_errors_are_loud(ops, cuda, gen):
    from lbm_amd._C import LbmkError
    x = randbf(gen, 10, 24).to(cuda)
    w = randbf(gen, 16, 24)
    pw = ops.pack_linear_weight(w.to(cuda), None)
    pw.Kp = 20  # corrupt: row stride smaller than K and not a multiple of 8
    with pytest.raises(LbmkError):
        ops.linear(x, pw)
    with pytest.raises(RuntimeError):
        ops.silu(torch.zeros(8, dtype=BF))  # CPU tensor: no fallback


@pytest.mark.parametrize("tile", [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19])
def test_conv3x3_wide_tiles(ops, cuda, gen, tile):
    """double-buffered 8-wave tiles (256x256 / 256x128) on a convolution with ragged M and N"""
    B, H, W, cin, cout = 2, 23, 19, 128, 320
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    res = randbf(gen, B, H, W, cout)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=tile)
    ref = _conv_ref(x, w, b, 1, 1) + res.float()
    assert rel_l2(y, ref) < 4e-3
    if tile in (17, 18, 19):      # the ping-pong tiles take 1x1 and plain 3x3 / stride-1 layers only
        from lbm_amd._C import LbmkError
        with pytest.raises(LbmkError):
            ops.conv2d(x.to(cuda), pw, up2=True, tile=tile)
        return
    y2 = ops.conv2d(x.to(cuda), pw, up2=True, tile=tile)
    assert rel_l2(y2, _conv_ref(x, w, b, 1, 1, up2=True)) < 4e-3


@pytest.mark.parametrize("tile", [-1, 0, 7, 8, 11, 14, 17, 18, 19])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 24, 128, 256), (1, 9, 7, 64, 96), (3, 8, 8, 192, 128)])
def test_conv_up2_as_four_parity_convolutions(ops, cuda, gen, tile, B, H, W, cin, cout):
    """nearest x2 upsample + 3x3 convolution computed as four 2x2 convolutions of the low-resolution input (4/9 of the
    products): against torch's upsample + conv2d in f32, and against the kernel that gathers through the upsample"""
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    parts = ops.pack_conv_weight_up2(w.to(cuda), b.to(cuda))
    assert [(pt, pl) for _, pt, pl in parts] == [(1, 1), (1, 0), (0, 1), (0, 0)]
    y = ops.conv2d_up2(x.to(cuda), parts, tile=tile)
    ref = _conv_ref(x, w, b, 1, 1, up2=True)
    assert y.shape == (B, 2 * H, 2 * W, cout)
    assert rel_l2(y, ref) < 4e-3
    assert (y.float().cpu() - ref).abs().max() < 0.1
    y9 = ops.conv2d(x.to(cuda), ops.pack_conv_weight(w.to(cuda), b.to(cuda)), up2=True)
    assert rel_l2(y, y9.float().cpu()) < 4e-3      # summed taps are rounded to bf16 once: not the same bits


def test_conv_up2_parity_convolutions_carry_groupnorm_stats(ops, cuda, gen):
    B, H, W, cin, cout, G = 2, 16, 32, 128, 256, 32
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    gam, bet = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
    parts = ops.pack_conv_weight_up2(w.to(cuda), b.to(cuda))
    for tile in (-1, 7, 17):
        y = ops.conv2d_up2(x.to(cuda), parts, gn_groups=G, tile=tile)
        assert getattr(y, "_lbmk_gn_partial", None) is not None
        y_plain = ops.conv2d_up2(x.to(cuda), parts, tile=tile)
        assert torch.equal(y, y_plain)
        n_fused = ops.groupnorm(y, gam.to(cuda), bet.to(cuda), G, 1e-6, True)
        ref = F.silu(F.group_norm(y.float().cpu().permute(0, 3, 1, 2), G, gam, bet, 1e-6)).permute(0, 2, 3, 1)
        assert rel_l2(n_fused, ref) < 4e-3
        n_plain = ops.groupnorm(y_plain, gam.to(cuda), bet.to(cuda), G, 1e-6, True)
        assert (n_fused.float() - n_plain.float()).abs().max() < 0.04


def test_conv3x3_every_tile_returns_the_same_bits(ops, cuda, gen):
    """every tile shape sums a layer's products in the same order (64-channel chunks, taps inside a chunk, two 32-deep
    MFMA steps per K-tile), so the automatic choice of a tile - which depends on the batch size - cannot change a bit
    of the result"""
    B, H, W, cin, cout = 2, 40, 36, 128, 320
    x = randbf(gen, B, H, W, cin).to(cuda)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    pw = ops.pack_conv_weight(w.to(cuda), randbf(gen, cout).to(cuda))
    res = randbf(gen, B, H, W, cout).to(cuda)
    ref = ops.conv2d(x, pw, residual=res, act=ops.ACT_SILU, tile=0)
    for tile in (1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, -1):
        assert torch.equal(ops.conv2d(x, pw, residual=res, act=ops.ACT_SILU, tile=tile), ref), tile


@pytest.mark.parametrize("tile", [7, 8, 11, 16, 17, 18, 19])
def test_conv3x3_many_tiles_two_images(ops, cuda, gen, tile):
    """16-wave tiles on a grid larger than the CU count, output tiles straddling the image boundary"""
    B, H, W, cin, cout = 2, 160, 161, 64, 320
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    res = randbf(gen, B, H, W, cout)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=tile)
    ref = _conv_ref(x, w, b, 1, 1) + res.float()
    assert rel_l2(y, ref) < 4e-3
    assert (y.float().cpu() - ref).abs().max() < 0.1


@pytest.mark.parametrize("tile", [15])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 23, 19, 128, 320), (2, 160, 161, 64, 320), (3, 9, 300, 64, 128),
                                             (5, 8, 8, 192, 96), (1, 40, 1, 64, 64), (1, 1, 70, 64, 64)])
def test_conv3x3_shared_activation_rows(ops, cuda, gen, tile, B, H, W, cin, cout):
    """tiles that stage each (filter row, channel chunk) of the input once for its three taps: ragged M and N,
    tiles straddling image rows and image boundaries, images narrower than a wave strip, one-pixel-wide / high images"""
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    res = randbf(gen, B, H, W, cout)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=tile)
    ref = _conv_ref(x, w, b, 1, 1) + res.float()
    assert rel_l2(y, ref) < 4e-3
    assert (y.float().cpu() - ref).abs().max() < 0.1
    y7 = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=7)
    assert torch.equal(y, y7)     # same products summed in the same order (chunk-major weights): the same bits


@pytest.mark.parametrize("tile", [15])
def test_conv3x3_shared_activation_rows_channel_window(ops, cuda, gen, tile):
    """input read as a channel window of a wider tensor (concatenated skip connections)"""
    B, H, W, cs, cin, cout = 2, 20, 24, 192, 64, 160
    x = randbf(gen, B, H, W, cs)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    pw = ops.pack_conv_weight(w.to(cuda), None)
    y = ops.conv2d(x.to(cuda), pw, cin_off=64, tile=tile)
    ref = _conv_ref(x[..., 64:128].contiguous(), w, None, 1, 1)
    assert rel_l2(y, ref) < 4e-3


def test_conv3x3_shared_activation_rows_rejects_other_shapes(ops, cuda, gen):
    from lbm_amd._C import LbmkError
    x = randbf(gen, 1, 16, 16, 64).to(cuda)
    pw = ops.pack_conv_weight(randbf(gen, 64, 64, 3, 3).to(cuda), None)
    with pytest.raises(LbmkError):
        ops.conv2d(x, pw, stride=2, tile=15)
    with pytest.raises(LbmkError):
        ops.conv2d(x, pw, up2=True, tile=15)


@pytest.mark.parametrize("cout,tile", [(128, -1), (128, 8), (128, 10), (256, 7), (512, 7), (256, 0), (256, 14), (128, 15), (256, 15), (512, 15), (256, 17), (512, 17), (128, 19)])
def test_conv_fused_groupnorm_stats(ops, cuda, gen, cout, tile):
    """the convolution's epilogue reduces the statistics of the GroupNorm that follows: same result as
    the stand-alone statistics pass, and equal to torch's group_norm on the stored tensor"""
    B, H, W, cin, G = 3, 24, 32, 64, 32
    x = randbf(gen, B, H, W, cin)
    w = randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    b = randbf(gen, cout)
    res = randbf(gen, B, H, W, cout)
    gam, bet = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
    pw = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    y = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=tile, gn_groups=G)
    assert getattr(y, "_lbmk_gn_partial", None) is not None
    y_plain = ops.conv2d(x.to(cuda), pw, residual=res.to(cuda), tile=tile)
    assert torch.equal(y, y_plain)
    n_fused = ops.groupnorm(y, gam.to(cuda), bet.to(cuda), G, 1e-6, True)
    n_plain = ops.groupnorm(y_plain, gam.to(cuda), bet.to(cuda), G, 1e-6, True)
    ref = F.silu(F.group_norm(y.float().cpu().permute(0, 3, 1, 2), G, gam, bet, 1e-6)).permute(0, 2, 3, 1)
    assert rel_l2(n_fused, ref) < 4e-3
    assert (n_fused.float() - n_plain.float()).abs().max() < 0.04   # one bf16 ulp at |x| <= 4


def test_conv_fused_groupnorm_stats_ineligible(ops, cuda, gen):
    """shapes the epilogue cannot reduce (10 channels per group) silently use the separate pass"""
    x = randbf(gen, 2, 16, 16, 64)
    w = randbf(gen, 320, 64, 3, 3, scale=(9 * 64) ** -0.5)
    pw = ops.pack_conv_weight(w.to(cuda), None)
    y = ops.conv2d(x.to(cuda), pw, gn_groups=32)
    assert getattr(y, "_lbmk_gn_partial", None) is None


@pytest.mark.parametrize("D,H,N", [(8, 32, 1024), (16, 4, 300), (32, 3, 520)])
def test_flash_attn_small_head_dims(ops, cuda, gen, D, H, N):
    """head dims below one 16-deep MFMA step / one 32-row output tile (pixel-space UNet mid block: 32 heads of 8)"""
    B = 2
    q, k, v = (randbf(gen, B, N, H * D) for _ in range(3))
    Np = (N + 7) // 8 * 8
    vt = torch.zeros(B, H * D, Np, dtype=BF)
    vt[:, :, :N] = v.transpose(1, 2)
    o = ops.flash_attn(q.to(cuda), k.to(cuda), vt.to(cuda), H, D)
    def heads(t):
        return t.float().view(B, N, H, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, N, H * D)
    assert rel_l2(o, ref) < 6e-3


@pytest.mark.parametrize("N,Cin,k", [(64, 3, 3), (4, 128, 3), (320, 320, 3), (640, 320, 1), (128, 192, 3), (24, 40, 3)])
def test_pack_conv_weight_kernel_matches_host_packing(ops, cuda, gen, N, Cin, k):
    """device packing (forward operand and input-gradient filter) is bit-identical to the torch-side re-layout"""
    w = torch.randn(N, Cin, k, k, generator=gen).bfloat16()
    b = torch.randn(N, generator=gen).bfloat16()
    ref = ops.pack_conv_weight(w, b)                                   # CPU tensors: host route
    dev = ops.pack_conv_weight(w.to(cuda), b.to(cuda))
    assert (dev.N, dev.K, dev.Kp, dev.Cin, dev.k_order, dev.alg_nk) == (ref.N, ref.K, ref.Kp, ref.Cin, ref.k_order, ref.alg_nk)
    assert torch.equal(dev.w.cpu(), ref.w) and torch.equal(dev.bias.cpu(), ref.bias)
    refd = ops.pack_conv_weight(w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), None)
    devd = ops.pack_conv_weight_dgrad(w.to(cuda))
    assert (devd.N, devd.K, devd.Kp, devd.Cin, devd.k_order, devd.alg_nk) == (refd.N, refd.K, refd.Kp, refd.Cin, refd.k_order, refd.alg_nk)
    assert torch.equal(devd.w.cpu(), refd.w)


@pytest.mark.parametrize("B,N,C", [(2, 1024, 512), (1, 328, 128), (3, 200, 64)])
def test_attention_two_pass(ops, cuda, gen, B, N, C):
    """single wide head: scores GEMM twice (row maxima, then bf16 exponentials + row sums), P V, normalise - against
    softmax(q k^T / sqrt(C)) v in f32, and against the unfused path with f32 scores"""
    q = randbf(gen, B, N, C)
    k = randbf(gen, B, N, C)
    v = randbf(gen, B, N, C)
    scale = C ** -0.5
    ref = torch.softmax(torch.einsum("bnc,bmc->bnm", q.float(), k.float()) * scale, -1) @ v.float()
    vt = v.transpose(1, 2).contiguous().to(cuda)
    o = ops.attention_two_pass(q.to(cuda), k.to(cuda), vt, scale)
    assert o.shape == (B, N, C) and o.dtype == BF
    assert rel_l2(o, ref) < 6e-3
    s = ops.gemm_nt(q.to(cuda), k.to(cuda), out_dtype=torch.float32, alpha=scale)
    o1 = ops.gemm_nt(ops.softmax_rows(s), vt)
    assert rel_l2(o, o1.float().cpu()) < 6e-3
    # a row with one dominant key: the maximum's own exponential is exactly 1
    qq = q.clone()
    qq[0, 5] = (k[0, 17].float() * 4).to(BF)
    o2 = ops.attention_two_pass(qq.to(cuda), k.to(cuda), vt, scale)
    ref2 = torch.softmax(torch.einsum("bnc,bmc->bnm", qq.float(), k.float()) * scale, -1) @ v.float()
    assert torch.isfinite(o2.float()).all() and rel_l2(o2, ref2) < 6e-3
