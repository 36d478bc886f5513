"""Backward (training) kernels against torch autograd in fp32 on the CPU, same seeded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def gen():
    return torch.Generator().manual_seed(1234)


def randbf(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(BF)


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def test_transpose_colsum(cuda, gen):
    from lbm_amd import ops
    x = randbf(gen, 136, 72)
    assert torch.equal(ops.transpose(x.to(cuda)).cpu(), x.t().contiguous())
    s = ops.colsum(x.to(cuda))
    assert rel_l2(s, x.float().sum(0, keepdim=True)) < 1e-5
    s4 = ops.colsum(x.to(cuda), 34, out_dtype=BF)
    assert rel_l2(s4, x.float().reshape(4, 34, 72).sum(1)) < 4e-3


# the last two take the split-K route (few rows, deep K) in the forward GEMM / in the input-gradient GEMM
@pytest.mark.parametrize("M,K,N", [(300 * 8, 320, 640), (16, 1280, 320), (4096, 64, 72), (520, 3840, 328), (264, 320, 4096)])
def test_linear_backward(cuda, gen, M, K, N):
    from lbm_amd import ops
    from lbm_amd.models.layers import HipLinear
    assert bool(ops.splitk_chunks(M, N, K)) == (K >= 3840) and bool(ops.splitk_chunks(M, K, N)) == (N >= 3840)
    lin = HipLinear(K, N).to(BF)
    with torch.no_grad():
        lin.weight.copy_(randbf(gen, N, K, scale=K ** -0.5))
        lin.bias.copy_(randbf(gen, N))
    x = randbf(gen, M, K)
    res = randbf(gen, M, N)
    dy = randbf(gen, M, N)
    # reference: fp32 autograd on the same bf16-rounded values
    xr, wr, br, rr = (t.float().requires_grad_() for t in (x, lin.weight.detach(), lin.bias.detach(), res))
    (F.linear(xr, wr, br) + rr).backward(dy.float())
    lin = lin.to(cuda)
    xg, rg = x.to(cuda).requires_grad_(), res.to(cuda).requires_grad_()
    y = lin(xg, residual=rg)
    y.backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 5e-3
    assert rel_l2(lin.weight.grad, wr.grad) < 5e-3
    assert rel_l2(lin.bias.grad, br.grad) < 5e-3
    assert torch.equal(rg.grad.cpu(), dy)


@pytest.mark.parametrize("cin,cout,H,W,stride,up2,asym", [(64, 128, 16, 24, 1, False, False), (320, 320, 16, 16, 2, False, False),
                                                          (64, 64, 8, 8, 1, True, False), (128, 128, 16, 16, 2, False, True),
                                                          (8, 320, 16, 16, 1, False, False), (320, 8, 16, 16, 1, False, False)])
def test_conv_backward(cuda, gen, cin, cout, H, W, stride, up2, asym):
    from lbm_amd.models.layers import HipConv2d
    B = 2
    conv = HipConv2d(cin, cout, 3, stride=stride, padding=0 if asym else 1, asym_pad=asym).to(BF)
    with torch.no_grad():
        conv.weight.copy_(randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5))
        conv.bias.copy_(randbf(gen, cout))
    x = randbf(gen, B, H, W, cin)
    # reference
    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    wr, br = conv.weight.detach().float().requires_grad_(), conv.bias.detach().float().requires_grad_()
    xi = F.interpolate(xr, scale_factor=2.0, mode="nearest") if up2 else xr
    if asym:
        yr = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wr, br, stride=2)
    else:
        yr = F.conv2d(xi, wr, br, stride=stride, padding=1)
    dy = randbf(gen, *yr.permute(0, 2, 3, 1).shape)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    conv = conv.to(cuda)
    xg = x.to(cuda).requires_grad_()
    y = conv(xg, up2=up2)
    assert y.shape[:3] == dy.shape[:3]
    dyp = F.pad(dy, (0, y.shape[-1] - dy.shape[-1])).to(cuda)     # thin outputs carry zero channels up to 8
    assert rel_l2(y[..., :cout], yr.permute(0, 2, 3, 1)) < 5e-3
    y.backward(dyp)
    assert rel_l2(xg.grad, xr.grad.permute(0, 2, 3, 1)) < 6e-3
    assert rel_l2(conv.weight.grad, wr.grad) < 6e-3
    assert rel_l2(conv.bias.grad, br.grad) < 6e-3


def test_conv_backward_rowvec_residual(cuda, gen):
    from lbm_amd.models.layers import HipConv2d
    B, H, W, cin, cout = 3, 8, 8, 64, 128
    conv = HipConv2d(cin, cout, 3, padding=1).to(BF)
    with torch.no_grad():
        conv.weight.copy_(randbf(gen, cout, cin, 3, 3, scale=(9 * cin) ** -0.5))
        conv.bias.copy_(randbf(gen, cout))
    x, rv, res = randbf(gen, B, H, W, cin), randbf(gen, B, cout), randbf(gen, B, H, W, cout)
    dy = randbf(gen, B, H, W, cout)
    xr, rvr, rr = x.float().requires_grad_(), rv.float().requires_grad_(), res.float().requires_grad_()
    yr = F.conv2d(xr.permute(0, 3, 1, 2), conv.weight.detach().float(), conv.bias.detach().float(), padding=1).permute(0, 2, 3, 1)
    (yr + rvr[:, None, None, :] + rr).backward(dy.float())
    conv = conv.to(cuda)
    xg, rvg, rg = x.to(cuda).requires_grad_(), rv.to(cuda).requires_grad_(), res.to(cuda).requires_grad_()
    conv(xg, rowvec=rvg, residual=rg).backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 6e-3
    assert rel_l2(rvg.grad, rvr.grad) < 6e-3
    assert torch.equal(rg.grad.cpu(), dy)


@pytest.mark.parametrize("B,H,W,C,silu", [(2, 16, 16, 320, True), (3, 8, 8, 1280, True), (2, 12, 20, 64, False), (2, 32, 32, 2560, True)])
def test_groupnorm_backward(cuda, gen, B, H, W, C, silu):
    from lbm_amd.autograd import GroupNormFn
    x, dy = randbf(gen, B, H, W, C), randbf(gen, B, H, W, C)
    gam, bet = (torch.rand(C, generator=gen) + 0.5).to(BF), torch.randn(C, generator=gen).to(BF)
    xr, gr, br = x.float().requires_grad_(), gam.float().requires_grad_(), bet.float().requires_grad_()
    yr = F.group_norm(xr.permute(0, 3, 1, 2), 32, gr, br, 1e-5)
    yr = (F.silu(yr) if silu else yr).permute(0, 2, 3, 1)
    yr.backward(dy.float())
    xg, gg, bg = x.to(cuda).requires_grad_(), gam.to(cuda).requires_grad_(), bet.to(cuda).requires_grad_()
    y = GroupNormFn.apply(xg, gg, bg, 32, 1e-5, silu)
    assert rel_l2(y, yr) < 5e-3
    y.backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 6e-3
    assert rel_l2(gg.grad, gr.grad) < 8e-3
    assert rel_l2(bg.grad, br.grad) < 8e-3


@pytest.mark.parametrize("rows,C", [(1000, 640), (512, 1280), (77, 320)])
def test_layernorm_backward(cuda, gen, rows, C):
    from lbm_amd.autograd import LayerNormFn
    x, dy = randbf(gen, rows, C), randbf(gen, rows, C)
    gam, bet = (torch.rand(C, generator=gen) + 0.5).to(BF), torch.randn(C, generator=gen).to(BF)
    xr, gr, br = x.float().requires_grad_(), gam.float().requires_grad_(), bet.float().requires_grad_()
    F.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy.float())
    xg, gg, bg = x.to(cuda).requires_grad_(), gam.to(cuda).requires_grad_(), bet.to(cuda).requires_grad_()
    LayerNormFn.apply(xg, gg, bg, 1e-5).backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 6e-3
    assert rel_l2(gg.grad, gr.grad) < 8e-3
    assert rel_l2(bg.grad, br.grad) < 8e-3


def test_silu_geglu_backward(cuda, gen):
    from lbm_amd.autograd import GegluFn, SiluFn
    x, dy = randbf(gen, 64, 320), randbf(gen, 64, 320)
    xr = x.float().requires_grad_()
    F.silu(xr).backward(dy.float())
    xg = x.to(cuda).requires_grad_()
    SiluFn.apply(xg).backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 5e-3
    h, dyh = randbf(gen, 50, 2 * 1280), randbf(gen, 50, 1280)
    hr = h.float().requires_grad_()
    a, g = hr.chunk(2, -1)
    yr = a * F.gelu(g)
    yr.backward(dyh.float())
    hg = h.to(cuda).requires_grad_()
    y = GegluFn.apply(hg)
    assert rel_l2(y, yr) < 5e-3
    y.backward(dyh.to(cuda))
    assert rel_l2(hg.grad, hr.grad) < 5e-3


# token counts that are multiples of 64: fused flash backward; (2, 72, 40, ...): recomputed scores; query counts that are
# not multiples of 8 (2x2 and 6x6 token grids of small training resolutions, with 5 lighting tokens as keys): zero-padded rows
@pytest.mark.parametrize("B,N,Nk,H,D", [(2, 256, 256, 5, 64), (1, 1024, 1024, 2, 64), (2, 384, 128, 3, 64),
                                        (2, 192, 192, 2, 64), (2, 64, 64, 4, 40), (1, 320, 192, 3, 40),
                                        (2, 256, 256, 2, 80), (1, 64, 128, 3, 80), (2, 128, 64, 2, 160),
                                        (1, 256, 256, 1, 160), (2, 72, 40, 2, 64), (2, 4, 4, 8, 160), (2, 36, 5, 2, 80),
                                        (1, 60, 64, 2, 64), (1, 36, 36, 1, 512)])
def test_attention_backward(cuda, gen, B, N, Nk, H, D):
    from lbm_amd.autograd import AttentionFn
    C = H * D
    q, do = randbf(gen, B, N, C), randbf(gen, B, N, C)
    k, v = randbf(gen, B, Nk, C), randbf(gen, B, Nk, C)
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    def heads(t):
        return t.view(B, -1, H, D).transpose(1, 2)
    orf = F.scaled_dot_product_attention(heads(qr), heads(kr), heads(vr)).transpose(1, 2).reshape(B, N, C)
    orf.backward(do.float())
    qg, kg, vg = (t.to(cuda).requires_grad_() for t in (q, k, v))
    o = AttentionFn.apply(qg, kg, vg, H, D, D ** -0.5)
    assert rel_l2(o, orf) < 6e-3
    o.backward(do.to(cuda))
    assert rel_l2(vg.grad, vr.grad) < 8e-3
    assert rel_l2(qg.grad, qr.grad) < 1e-2
    assert rel_l2(kg.grad, kr.grad) < 1e-2


def test_flash_backward_peaked_rows_and_reproducible(cuda, gen):
    """scores with a large dynamic range (rows dominated by one key) and two runs bit-identical"""
    from lbm_amd import ops
    B, N, H, D = 2, 256, 2, 64
    C = H * D
    q, k, v, do = (randbf(gen, B, N, C) for _ in range(4))
    q = (q * 6).to(torch.bfloat16)
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    def heads(t):
        return t.view(B, -1, H, D).transpose(1, 2)
    orf = F.scaled_dot_product_attention(heads(qr), heads(kr), heads(vr)).transpose(1, 2).reshape(B, N, C)
    orf.backward(do.float())
    qc, kc, vc, dc = (t.to(cuda) for t in (q, k, v, do))
    o, lse = ops.flash_attn(qc, kc, ops.transpose(vc), H, D, D ** -0.5, return_lse=True)
    sc = torch.einsum("bhqd,bhkd->bhqk", heads(q.float()), heads(k.float())) * D ** -0.5
    assert (lse.cpu() - torch.logsumexp(sc, -1) * 1.4426950408889634).abs().max() < 2e-2
    g1 = ops.flash_attn_bwd(qc, kc, vc, o, dc, lse, H, D, D ** -0.5)
    g2 = ops.flash_attn_bwd(qc, kc, vc, o, dc, lse, H, D, D ** -0.5)
    for a, b_, ref, tol in zip(g1, g2, (qr.grad, kr.grad, vr.grad), (2e-2, 2e-2, 1e-2)):
        assert torch.equal(a, b_)
        assert rel_l2(a, ref) < tol


@pytest.mark.parametrize("flat", [False, True])
def test_self_attention_fused_qkv_projection(cuda, gen, flat):
    """one GEMM for the three projections (stacked weights; a view when they are adjacent in a flat buffer)"""
    from lbm_amd.models.unets.unet2d_condition import Attention
    B, N, H, D = 2, 128, 3, 64
    C = H * D
    torch.manual_seed(3)
    att = Attention(C, H, D, None).to(torch.bfloat16)
    x, res, dy = randbf(gen, B, N, C), randbf(gen, B, N, C), randbf(gen, B, N, C)
    ws = {n: p.detach().float().requires_grad_() for n, p in att.named_parameters()}
    xr = x.float().requires_grad_()
    def heads(t):
        return t.view(B, N, H, D).transpose(1, 2)
    q, k, v = (xr @ ws[f"to_{r}.weight"].t() for r in "qkv")
    orf = F.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, N, C)
    yr = orf @ ws["to_out.0.weight"].t() + ws["to_out.0.bias"] + res.float()
    yr.backward(dy.float())
    att = att.to(cuda)
    if flat:      # re-home the parameters back to back, in reverse order, as the trainer's flat buffers do
        ps = [att.to_v.weight, att.to_k.weight, att.to_q.weight]
        buf = torch.cat([p.data.reshape(-1) for p in ps])
        o = 0
        for p in ps:
            p.data = buf[o:o + p.numel()].view_as(p)
            o += p.numel()
        stack, cols = att._qkv_stack()
        assert stack.data_ptr() == buf.data_ptr() and cols == {"v": 0, "k": C, "q": 2 * C}
    xg = x.to(cuda).requires_grad_()
    y = att(xg, None, res.to(cuda))
    assert rel_l2(y, yr) < 8e-3
    y.backward(dy.to(cuda))
    assert rel_l2(xg.grad, xr.grad) < 1.5e-2
    for n, p in att.named_parameters():
        assert rel_l2(p.grad, ws[n].grad) < 1.5e-2, n


@pytest.mark.parametrize("l1", [False, True])
def test_latent_loss_backward(cuda, gen, l1):
    from lbm_amd.autograd import LatentLossFn
    B = 3
    pred, zs, z = randbf(gen, B, 4, 16, 16), randbf(gen, B, 4, 16, 16), randbf(gen, B, 4, 16, 16)
    mask = (torch.rand(B, 1, 16, 16, generator=gen) > 0.3)
    w = torch.tensor([0.5, 1.0, 2.0])
    pr = pred.float().requires_grad_()
    tgt = (zs - z).float()
    m = mask.float()
    d = pr * m - tgt * m
    per = (d.abs() if l1 else d ** 2).mean(dim=(1, 2, 3))
    (per * w).sum().backward()
    pg = pred.to(cuda).requires_grad_()
    out = LatentLossFn.apply(pg, zs.to(cuda), z.to(cuda), mask.to(cuda), l1)
    assert rel_l2(out, per) < 1e-3
    (out * w.to(cuda)).sum().backward()
    assert rel_l2(pg.grad, pr.grad) < 5e-3


SMALL_UNET = dict(
    in_channels=4, out_channels=4,
    down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
    block_out_channels=[64, 128, 256], layers_per_block=2, cross_attention_dim=[64, 128, 256],
    transformer_layers_per_block=[1, 1, 2], attention_head_dim=[1, 2, 4], use_linear_projection=True)


def test_unet_gradients_match_oracle_autograd(cuda):
    """Every parameter gradient of a small SDXL-style UNet (down/up sampling, skip concatenation, time
    embedding, self-attention, GEGLU) against fp32 autograd through the CPU restatement of the same UNet."""
    from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
    from oracle import diffusers_ref as dr
    torch.manual_seed(0)
    m = DiffusersUNet2DCondWrapper(**SMALL_UNET).to(BF)
    sd = {k: v.detach().float().cpu().requires_grad_() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    B, L = 2, 32
    x = torch.randn(B, 4, L, L, generator=g).to(BF)
    ts = torch.tensor([250.0, 750.0])
    w = torch.randn(B, 4, L, L, generator=g).to(BF)
    ref = dr.unet_forward(sd, SMALL_UNET, x.float(), ts)
    (ref * w.float()).sum().backward()

    m = m.to(cuda).train()
    out = m(sample=x.to(cuda), timestep=ts.to(cuda), conditioning={"cond": {}})
    assert out.requires_grad and rel_l2(out, ref) < 1e-2
    (out.float() * w.to(cuda).float()).sum().backward()

    num = den = 0.0
    worst = (0.0, None)
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        gr = sd[name].grad
        a, b = p.grad.detach().float().cpu(), gr
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
        if b.numel() >= 4096:
            cos = float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30))
            if 1 - cos > worst[0]:
                worst = (1 - cos, name)
    total = (num / den) ** 0.5
    print(f"global grad rel-L2 {total:.4f}; worst 1-cos {worst[0]:.5f} at {worst[1]}")
    assert total < 5e-2
    assert worst[0] < 1e-2


def test_adamw_kernel_matches_torch(cuda, gen):
    from lbm_amd import ops
    n = 1000 * 8 + 3
    p0 = randbf(gen, n)
    ref_p = p0.float().clone().requires_grad_()
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p = torch.zeros(n + 5, dtype=BF, device=cuda)[:n]
    p.copy_(p0.to(cuda))
    m = torch.zeros(n, device=cuda)
    v = torch.zeros(n, device=cuda)
    for step in range(1, 4):
        g = randbf(gen, n)
        ref_p.grad = g.float() * 0.5
        opt.step()
        ops.adamw_step(p, g.to(cuda), m, v, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, step=step, grad_scale=0.5)
    # bf16 storage of the parameter after each step bounds the agreement
    assert rel_l2(p, ref_p) < 5e-3
    assert rel_l2(m, opt.state[ref_p]["exp_avg"]) < 1e-5
    assert rel_l2(v, opt.state[ref_p]["exp_avg_sq"]) < 1e-5


def _toy_lbm(cuda, loss_type="l2"):
    from lbm_amd.models.embedders import ConditionerWrapper
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    from lbm_amd.models.schedulers import FlowMatchEulerDiscreteScheduler
    from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
    torch.manual_seed(0)
    den = DiffusersUNet2DCondWrapper(**SMALL_UNET).to(BF)
    cfg = LBMConfig(source_key="source", target_key="target", latent_loss_type=loss_type, latent_loss_weight=1.0,
                    pixel_loss_weight=0.0, timestep_sampling="custom_timesteps", selected_timesteps=[250, 500, 750, 1000],
                    prob=[0.25, 0.25, 0.25, 0.25], bridge_noise_sigma=0.005)
    model = LBMModel(cfg, denoiser=den, training_noise_scheduler=FlowMatchEulerDiscreteScheduler(),
                     sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler(), vae=None,
                     conditioner=ConditionerWrapper([])).to(BF).to(cuda)
    return model


def test_lbm_training_step_against_oracle(cuda):
    """LBMModel.forward in training mode: loss and denoiser gradients against the fp32 oracle
    (bridge restatement + UNet restatement under torch autograd) on the same latents, noise and timesteps."""
    from oracle import bridge_ref as br
    from oracle import diffusers_ref as dr
    model = _toy_lbm(cuda)
    sd = {k: v.detach().float().cpu().requires_grad_() for k, v in model.denoiser.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    B, L = 2, 32
    z, zs, eps = (torch.randn(B, 4, L, L, generator=g).to(BF) for _ in range(3))
    ts = torch.tensor([250, 1000])
    # oracle
    sigmas = (ts.float() / 1000).view(B, 1, 1, 1)
    x_t = br.interpolant(zs.float(), z.float(), sigmas, eps.float(), 0.005)
    x_t[1] = zs[1].float()                                   # t == timesteps[0]: exact source
    pred = dr.unet_forward(sd, SMALL_UNET, x_t, ts.float())
    ref_loss = ((pred - (zs - z).float()) ** 2).mean(dim=(1, 2, 3)).mean()
    ref_loss.backward()

    model.train()
    out = model({"target": z.to(cuda), "source": zs.to(cuda)}, bridge_eps=eps.to(cuda), timesteps=ts.to(cuda))
    assert abs(float(out["loss"].detach()) - float(ref_loss.detach())) < 2e-2 * float(ref_loss.detach())
    out["loss"].backward()
    num = den = 0.0
    for name, p in model.denoiser.named_parameters():
        a, b = p.grad.detach().float().cpu(), sd[name].grad
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    assert (num / den) ** 0.5 < 6e-2


def test_training_pipeline_reduces_loss(cuda):
    """fit_step (flat parameter / gradient buffers, fused AdamW) on a fixed batch: the loss goes down, every
    trainable parameter moves, parameters stay views of the flat buffer."""
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    model = _toy_lbm(cuda)
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=2e-4, optimizer_name="AdamW", trainable_params=["denoiser.*"],
                                                  optimizer_kwargs={"weight_decay": 0.0}))
    pipe.configure_optimizers()
    g = torch.Generator().manual_seed(3)
    z, zs, eps = (torch.randn(2, 4, 32, 32, generator=g).to(BF).to(cuda) for _ in range(3))
    ts = torch.tensor([500, 750], device=cuda)
    before = {n: p.detach().clone() for n, p in model.denoiser.named_parameters()}
    losses = []
    orig_forward = model.forward
    model.forward = lambda batch, *a, **k: orig_forward(batch, bridge_eps=eps, timesteps=ts)
    for _ in range(6):
        losses.append(float(pipe.fit_step({"target": z, "source": zs})["loss"]))
    print("losses", losses)
    assert losses[-1] < 0.9 * losses[0]
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.denoiser.named_parameters())
    # bf16 parameters: a 2e-4 step is below half an ulp of the norm scales initialised at 1.0, those stay put
    assert moved > 0.8 * len(before)
    grp = pipe.optims[0].groups[0]
    lo, hi = grp.flat.data_ptr(), grp.flat.data_ptr() + grp.flat.numel() * 2
    assert all(lo <= p.data_ptr() < hi for p in grp.params)


def test_direct_gradient_route_matches_autograd_accumulation(cuda):
    """inside a trainer step, weight gradients are written straight into the flat gradient buffer (the stacked
    q/k/v block in one GEMM); the result equals what autograd's accumulation of returned gradients gives"""
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    model = _toy_lbm(cuda)
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=1e-4, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
    pipe.configure_optimizers()
    g = torch.Generator().manual_seed(5)
    z, zs, eps = (torch.randn(2, 4, 32, 32, generator=g).to(BF).to(cuda) for _ in range(3))
    ts = torch.tensor([500, 750], device=cuda)
    grp = pipe.optims[0].groups[0]
    model.train()
    # direct route
    pipe.reducer.begin_step()
    model({"target": z, "source": zs}, bridge_eps=eps, timesteps=ts)["loss"].backward()
    pipe.reducer.finish()
    direct = grp.grad.clone()
    n_direct = len(pipe.reducer.written)
    assert n_direct > 0.4 * len(grp.params)       # every convolution / linear weight took it (the rest: biases, norm affines)
    # plain autograd accumulation onto zeroed gradients (sink inactive outside a step)
    grp.grad.zero_()
    grp.rebind_grads()
    model({"target": z, "source": zs}, bridge_eps=eps, timesteps=ts)["loss"].backward()
    assert float(direct.float().abs().max()) > 0
    assert rel_l2(direct, grp.grad) < 1e-6


def test_pixel_loss_through_vae_decoder(cuda):
    """pixel loss (l2): gradient w.r.t. the latent through the frozen VAE decoder (incl. the single-head
    d=512 mid attention) against fp32 autograd through the oracle decoder."""
    from lbm_amd.autograd import PixelLossFn
    from lbm_amd.models.vae import AutoencoderKL, AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig
    from oracle import diffusers_ref as dr
    torch.manual_seed(1)
    core = AutoencoderKL(scaling_factor=0.13025).to(BF)
    sd = {k: v.detach().float().cpu() for k, v in core.state_dict().items()}
    vae = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version="unused", tiling_size=(128, 128)), vae_model=core).to(cuda)
    vae.freeze()
    g = torch.Generator().manual_seed(2)
    B, L = 2, 16
    z = (torch.randn(B, 4, L, L, generator=g) * 0.5).to(BF)
    tgt = (torch.rand(B, 3, 8 * L, 8 * L, generator=g) * 2 - 1).to(BF)
    zr = z.float().requires_grad_()
    dec_r = dr.vae_decode(sd, zr / 0.13025).clamp(-1, 1)
    per_r = ((dec_r - tgt.float()) ** 2).mean(dim=(1, 2, 3))
    per_r.sum().backward()
    zg = z.to(cuda).float().requires_grad_()
    dec = vae.decode(zg)
    per = PixelLossFn.apply(dec, tgt.to(cuda), None, False)
    assert rel_l2(per, per_r) < 2e-2
    per.sum().backward()
    assert zg.grad is not None and rel_l2(zg.grad, zr.grad) < 6e-2
    cos = float((zg.grad.cpu().double() * zr.grad.double()).sum() / (zg.grad.cpu().double().norm() * zr.grad.double().norm()))
    assert cos > 0.995


def test_lbm_forward_with_pixel_loss(cuda):
    """LBMModel.forward with pixel_loss_weight > 0 (l1): finite loss, gradients reach the denoiser."""
    from lbm_amd.models.vae import AutoencoderKL, AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig
    model = _toy_lbm(cuda)
    torch.manual_seed(1)
    core = AutoencoderKL(scaling_factor=0.13025).to(BF)
    model.vae = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version="unused", tiling_size=(128, 128)), vae_model=core).to(cuda)
    model.vae.freeze()
    model.pixel_loss_weight, model.pixel_loss_type, model.pixel_loss_max_size = 10.0, "l1", 64
    model.train()
    g = torch.Generator().manual_seed(5)
    imgs = {k: (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).to(BF).to(cuda) for k in ("target", "source")}
    out = model(imgs, timesteps=torch.tensor([500, 250], device=cuda))
    assert float(out["pixel_recon_loss"].detach()) > 0 and torch.isfinite(out["loss"].detach())
    assert float(out["loss"].detach()) > float(out["latent_recon_loss"].detach())
    out["loss"].backward()
    grads = [p.grad for p in model.denoiser.parameters()]
    assert all(gd is not None and torch.isfinite(gd.float()).all() for gd in grads)
    assert all(p.grad is None for p in model.vae.parameters())


def test_training_checkpoint_resume(cuda, tmp_path):
    """save after 2 steps, keep training 2 more; a fresh pipeline restored from the file and trained on the same
    2 batches ends bit-identical (parameters, moments, step), and get_model-style loading sees the weights."""
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    cfg = TrainingConfig(learning_rate=1e-3, optimizer_name="AdamW", trainable_params=["denoiser.*"])
    g = torch.Generator().manual_seed(3)
    z, zs, eps = (torch.randn(2, 4, 32, 32, generator=g).to(BF).to(cuda) for _ in range(3))
    ts = torch.tensor([500, 750], device=cuda)

    def make():
        model = _toy_lbm(cuda)
        orig = model.forward
        model.forward = lambda batch, *a, **k: orig(batch, bridge_eps=eps, timesteps=ts)
        pipe = TrainingPipeline(model, cfg)
        pipe.configure_optimizers()
        return model, pipe

    model, pipe = make()
    for _ in range(2):
        pipe.fit_step({"target": z, "source": zs})
    path = str(tmp_path / "last.ckpt")
    pipe.save_checkpoint(path)
    for _ in range(2):
        pipe.fit_step({"target": z, "source": zs})
    model2, pipe2 = make()
    pipe2.load_checkpoint(path)
    assert pipe2.global_step == 2 and pipe2.optims[0].step_count == 2
    for _ in range(2):
        pipe2.fit_step({"target": z, "source": zs})
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(p1, p2), n1
    assert torch.equal(pipe.optims[0].groups[0].m, pipe2.optims[0].groups[0].m)
    sd = torch.load(path, map_location="cpu", weights_only=True)["state_dict"]
    assert all(k.startswith("model.") for k in sd) and "model.denoiser.conv_in.weight" in sd


def test_frozen_input_layers_do_not_starve_the_rest(cuda):
    """a trainable_params pattern that freezes conv_in and the time-embedding MLP: the forward still takes the
    differentiable route, the remaining parameters (resnet time projections included) receive gradients"""
    model = _toy_lbm(cuda)
    den = model.denoiser
    for p in list(den.conv_in.parameters()) + list(den.time_embedding.parameters()):
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(2)
    z, zs, eps = (torch.randn(2, 4, 32, 32, generator=g).to(BF).to(cuda) for _ in range(3))
    out = model({"target": z, "source": zs}, bridge_eps=eps, timesteps=torch.tensor([500, 750], device=cuda))
    out["loss"].backward()
    proj = den.down_blocks[0].resnets[0].time_emb_proj.weight
    assert proj.grad is not None and float(proj.grad.float().abs().sum()) > 0
    assert den.conv_out.weight.grad is not None and den.conv_in.weight.grad is None


def test_fit_loop_resume_and_rccl_exchange(cuda, tmp_path):
    """lbm_amd.trainer.fit: 4 steps straight through == 2 steps, stop, resume from the checkpoint the resolver picks,
    2 more steps (bit-identical parameters); then the same 4 steps through a one-rank RCCL process group with the
    reduce-scatter -> sharded AdamW -> all-gather exchange give the same parameters as without a process group
    (with one rank the collectives are copies, so any difference would be a bookkeeping error)."""
    import socket
    import torch.distributed as dist
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline, fit, resolve_resume_checkpoint, resolve_task_dir
    cfg = TrainingConfig(learning_rate=1e-3, optimizer_name="AdamW", trainable_params=["denoiser.*"])
    g = torch.Generator().manual_seed(5)
    batches = [{"target": torch.randn(2, 4, 32, 32, generator=g).to(BF), "source": torch.randn(2, 4, 32, 32, generator=g).to(BF)}
               for _ in range(4)]
    eps = torch.randn(2, 4, 32, 32, generator=g).to(BF).to(cuda)
    ts = torch.tensor([500, 750], device=cuda)

    def make(d=None, **kw):
        model = _toy_lbm(cuda)
        orig = model.forward
        model.forward = lambda batch, *a, **k: orig(batch, bridge_eps=eps, timesteps=ts)
        pipe = TrainingPipeline(model, cfg, dist=d, **kw)
        pipe.configure_optimizers()
        return model, pipe

    to_dev = lambda b: {k: v.to(cuda) for k, v in b.items()}
    model_a, pipe_a = make()
    assert fit(pipe_a, batches, max_epochs=1, to_device=to_dev) == 4
    task = resolve_task_dir(str(tmp_path), "toy")
    model_b, pipe_b = make()
    fit(pipe_b, batches[:2], max_epochs=1, save_dir=task, save_interval=2, to_device=to_dev)
    ckpt = resolve_resume_checkpoint(task, True)
    assert ckpt is not None and ckpt.endswith("epoch=0-step=2.ckpt")
    model_c, pipe_c = make()
    assert fit(pipe_c, batches[2:], max_epochs=1, ckpt_path=ckpt, to_device=to_dev) == 4
    for (n1, p1), (_, p2) in zip(model_a.named_parameters(), model_c.named_parameters()):
        assert torch.equal(p1, p2), n1
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=cuda)
    try:
        model_d, pipe_d = make(dist, grad_exchange="reduce_scatter", bucket_bytes=1 << 16)
        assert len(pipe_d.optims[0].groups[0].buckets) > 3
        outs = [pipe_d.fit_step(to_dev(b)) for b in batches]
        torch.cuda.synchronize()
        assert abs(float(outs[-1]["loss_mean"]) - float(outs[-1]["loss"])) < 1e-6
    finally:
        dist.destroy_process_group()
    for (n1, p1), (_, p2) in zip(model_a.named_parameters(), model_d.named_parameters()):
        assert torch.equal(p1, p2), n1


@pytest.mark.parametrize("mode", ["relight_self_attn", "filllight_cross_attn"])
def test_unet_sd15_style_gradients(cuda, mode):
    """SD1.5-style UNet (1x1-conv projections, head dims 40 / 80): parameter gradients against fp32 autograd through
    the restatement; in the fill-light mode attn2 attends to 5 lighting tokens (keys padded to 8 and masked) and the
    gradient reaches the context (the lighting embedder is trained with the denoiser)."""
    from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
    from oracle import diffusers_ref as dr
    from test_models_gpu import SD15_STYLE
    cad = [320, 640] if mode == "relight_self_attn" else 768
    cfg = dict(SD15_STYLE, cross_attention_dim=cad)
    torch.manual_seed(4)
    m = DiffusersUNet2DCondWrapper(**cfg).to(BF)
    sd = {k: v.detach().float().cpu().requires_grad_() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 4, 16, 16, generator=g).to(BF)
    c = torch.randn(2, 8, 16, 16, generator=g).to(BF)
    w = torch.randn(2, 4, 16, 16, generator=g).to(BF)
    ts = torch.tensor([750.0, 250.0])
    cond = {"concat": c.to(cuda)}
    ctx_r = ctx_g = None
    if mode == "filllight_cross_attn":
        ctx = torch.randn(2, 5, 768, generator=g).to(BF)
        ctx_r = ctx.float().requires_grad_()
        ctx_g = ctx.to(cuda).requires_grad_()
        cond["crossattn"] = ctx_g
    ref = dr.unet_forward(sd, cfg, torch.cat([x, c], 1).float(), ts, ctx_r)
    (ref * w.float()).sum().backward()
    m = m.to(cuda).train()
    out = m(sample=x.to(cuda), timestep=ts.to(cuda), conditioning={"cond": cond})
    assert rel_l2(out, ref) < 1e-2
    (out.float() * w.to(cuda).float()).sum().backward()
    num = den = 0.0
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        a, b = p.grad.detach().float().cpu(), sd[name].grad
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    assert (num / den) ** 0.5 < 6e-2
    if ctx_g is not None:
        assert ctx_g.grad is not None and rel_l2(ctx_g.grad, ctx_r.grad) < 6e-2


def test_conv_weight_gradient_in_image_groups(cuda, gen):
    """the transposed operands of a convolution's weight gradient are addressed with 32-bit byte offsets: past 2 GiB
    (a 1024x1024 batch-8 training step) the images are processed in groups with f32 partial sums.  Forced here by
    lowering the window: same gradient as in one piece."""
    from lbm_amd import autograd
    from lbm_amd.models.layers import HipConv2d
    torch.manual_seed(3)
    conv = HipConv2d(64, 72, 3, padding=1).to(BF).to(cuda)
    x = randbf(gen, 5, 16, 16, 64).to(cuda).requires_grad_(True)
    dy = randbf(gen, 5, 16, 16, 72).to(cuda)

    def grads():
        conv.weight.grad = None
        conv(x).backward(dy)
        return conv.weight.grad.clone()

    whole = grads()
    old, autograd.WGRAD_WINDOW_BYTES = autograd.WGRAD_WINDOW_BYTES, 2 * 576 * 2 * 256      # two images per group (K9 = 576, 256 pixels each)
    try:
        grouped = grads()
    finally:
        autograd.WGRAD_WINDOW_BYTES = old
    ref = torch.nn.grad.conv2d_weight(x.detach().float().permute(0, 3, 1, 2), conv.weight.shape, dy.float().permute(0, 3, 1, 2), padding=1)
    assert rel_l2(whole, ref) < 6e-3 and rel_l2(grouped, ref) < 6e-3
    assert rel_l2(grouped, whole) < 6e-3 and not torch.equal(grouped, torch.zeros_like(grouped))


@pytest.mark.parametrize("M,N,K,splits", [(1000, 320, 200, None), (2048, 1280, 640, 1), (2048, 10240, 1280, None), (2048, 1280, 640, 3),
                                          (4104, 136, 72, None), (64, 128, 128, None), (8192, 2560, 640, None), (333, 1288, 24, 4)])
def test_gemm_nn_untransposed_weight(cuda, gen, M, N, K, splits):
    """dX = dY W straight from the parameter as stored ([N, K] row-major, transposing LDS reads for the weight tile):
    ragged tiles in M and K, reduction lengths that are not whole stages, forced and automatic splits of the reduction,
    a strided dY, and a fixed summation order"""
    from lbm_amd import ops
    a = torch.empty(M, N + 8, dtype=BF, device=cuda)[:, :N].copy_(randbf(gen, M, N).to(cuda))     # row stride N + 8
    w = (randbf(gen, N, K) * 0.1).to(BF).to(cuda)
    ref = a.float().cpu() @ w.float().cpu()
    assert ops.gemm_nn_ok(a, w)
    out = ops.gemm_nn(a, w, splits=splits)
    assert out.shape == (M, K) and out.dtype == BF and rel_l2(out, ref) < 4e-3
    assert torch.equal(ops.gemm_nn(a, w, splits=splits), out)
    if splits is None:
        # same product as the forward GEMM on a W^T copy (different summation order: equal to rounding)
        alt = ops.linear(a, ops.pack_linear_weight(w.t().contiguous(), None))
        assert rel_l2(out, alt) < 4e-3


@pytest.mark.parametrize("M,N,K,splits", [(1000, 320, 200, None), (2048, 1280, 640, 1), (2048, 1280, 640, 3), (4096, 136, 72, None),
                                          (64, 128, 128, None), (8192, 640, 2560, None), (333 * 8, 24, 1288, 4)])
def test_gemm_tn_untransposed_operands(cuda, gen, M, N, K, splits):
    """dW = dY^T X straight from the row-major operands (transposing LDS reads): ragged tiles in N and K, row counts
    that are not whole stages, forced split counts, strided operands, bf16 output into a preallocated slot"""
    from lbm_amd import ops
    a = randbf(gen, M, N + 8)[:, :N]            # row stride N + 8
    b = randbf(gen, M, K)
    ref = a.float().t() @ b.float()
    ag, bg = a.to(cuda), b.to(cuda)
    ag = torch.empty(M, N + 8, dtype=BF, device=cuda)[:, :N].copy_(ag)
    out = ops.gemm_tn(ag, bg, splits=splits)
    assert out.dtype == torch.float32 and rel_l2(out, ref) < 2e-3
    slot = torch.zeros(N * K + 16, dtype=BF, device=cuda)[8:8 + N * K].view(N, K)
    ops.gemm_tn(ag, bg, out=slot, splits=splits)
    assert rel_l2(slot, ref) < 4e-3
    assert torch.equal(ops.gemm_tn(ag, bg, splits=splits), out)             # fixed summation order
