"""LPIPS (VGG16) on the HIP kernels against the fp32 restatement in oracle/lpips_ref.py and torch primitives.
The lpips package and its weights are not available offline: weights here are seeded random (He-initialised
convolutions, non-negative heads), so this pins the arithmetic, not the published numbers (parity unpinned)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _randomise(m, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.startswith("net.") and n.endswith("weight"):
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[1] * 9)) ** 0.5)
            elif n.startswith("net."):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.rand(p.shape, generator=g) * 0.2)
    return m


def test_maxpool_and_relu_backward(cuda):
    from lbm_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 8, 12, generator=g).to(BF)          # NCHW on the torch side
    x[:, :4] = x[:, :4].clamp_min(0).round()                      # ties (and all-zero windows) in some channels
    dy = torch.randn(2, 16, 4, 6, generator=g).to(BF)
    xr = x.float().requires_grad_()
    yr = F.max_pool2d(xr, 2, 2)
    yr.backward(dy.float())
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    y = ops.maxpool2(xh)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2).float(), yr.detach())
    dx = ops.maxpool2_bwd(xh, dy.permute(0, 2, 3, 1).contiguous().to(cuda))
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2).float(), xr.grad)
    yv = torch.randn(4096, generator=g).to(BF)
    gv = torch.randn(4096, generator=g).to(BF)
    d = ops.relu_bwd(yv.to(cuda), gv.to(cuda)).cpu()
    assert torch.equal(d, torch.where(yv > 0, gv, torch.zeros_like(gv)))


@pytest.mark.parametrize("C,HW,dropout", [(64, 300, False), (128, 256, True), (256, 77, False), (512, 64, True)])
def test_lpips_head_forward_backward(cuda, C, HW, dropout):
    from lbm_amd import ops
    from oracle.lpips_ref import normalize_tensor
    g = torch.Generator().manual_seed(C)
    B = 3
    f0 = torch.randn(B, HW, C, generator=g).clamp_min(0).to(BF)
    f1 = torch.randn(B, HW, C, generator=g).clamp_min(0).to(BF)
    f0[0, 5] = 0                                                   # an all-zero pixel: finite gradient
    w = torch.rand(C, generator=g)
    keep = (torch.rand(B, HW, C, generator=g) >= 0.5) if dropout else None
    gout = torch.randn(B, generator=g)
    a = f0.float().requires_grad_()
    d = (normalize_tensor(a.transpose(1, 2)) - normalize_tensor(f1.float().transpose(1, 2))) ** 2      # [B, C, HW]
    if dropout:
        d = d * keep.transpose(1, 2) * 2.0
    ref = (d * w[None, :, None]).sum(1).mean(1)
    mask = torch.ones(B, HW, 1)
    mask[0, 5] = 0                                                 # torch's sqrt backward is NaN at the zero pixel
    out = ops.lpips_head(f0.to(cuda), f1.to(cuda), w.to(cuda), keep.to(torch.uint8).to(cuda) if dropout else None)
    assert rel_l2(out, ref) < 1e-5
    (ref * gout).sum().backward()
    df = ops.lpips_head_bwd(f0.to(cuda), f1.to(cuda), w.to(cuda), keep.to(torch.uint8).to(cuda) if dropout else None, gout.to(cuda))
    assert torch.isfinite(df.float()).all()
    assert rel_l2(df.cpu().float() * mask, torch.nan_to_num(a.grad) * mask) < 6e-3


def test_lpips_module_against_oracle(cuda):
    from lbm_amd.models.lpips import LPIPS
    from oracle import lpips_ref
    m = _randomise(LPIPS())
    sd = {k: v.detach().float().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    B, H = 2, 64
    in0 = (torch.rand(B, 3, H, H, generator=g) * 2.4 - 1.2)        # some values outside [-1, 1] for the clamp
    in1 = (torch.rand(B, 3, H, H, generator=g) * 2 - 1)
    mask = (torch.rand(B, 1, H, H, generator=g) > 0.2).float()
    a = in0.clone().requires_grad_()
    ref = lpips_ref.lpips_forward(sd, a.clamp(-1, 1) * mask, in1 * mask)
    ref.mean().backward()
    m = m.to(BF).to(cuda)
    assert not m.training                                           # eval_mode=True as in the package
    x = in0.to(cuda).requires_grad_()
    out = m(x, in1.to(cuda), mask=mask.to(cuda), clamp0=True)
    assert out.shape == (B, 1, 1, 1)
    assert rel_l2(out, ref) < 3e-2
    out.mean().backward()
    # the yardstick for the gradient is torch's own bf16 autocast (the reference's execution mode) against the same
    # fp32 oracle: 13 ReLU layers and a difference of unit-normalised features make it ~18 % here
    b = in0.clone().requires_grad_()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_bf = lpips_ref.lpips_forward(sd, b.clamp(-1, 1) * mask, in1 * mask)
    out_bf.float().mean().backward()
    err_torch_bf16 = rel_l2(b.grad, a.grad)
    assert rel_l2(x.grad, a.grad) < max(8e-2, 1.25 * err_torch_bf16)
    cos = float((x.grad.cpu().double() * a.grad.double()).sum() / (x.grad.cpu().double().norm() * a.grad.double().norm()))
    assert cos > 0.97
    # plain two-argument call (the package's signature), no gradient
    with torch.no_grad():
        out2 = m(in0.to(cuda).to(BF), in1.to(cuda).to(BF))
    assert rel_l2(out2, lpips_ref.lpips_forward(sd, in0.to(BF).float(), in1.to(BF).float())) < 3e-2
    # train mode: dropout in front of the heads is active (mean over draws stays near the eval value)
    m.train()
    with torch.no_grad():
        vals = torch.stack([m(in0.to(cuda), in1.to(cuda)) for _ in range(8)])
    assert float(vals.std()) > 0 and rel_l2(vals.mean(0), lpips_ref.lpips_forward(sd, in0, in1)) < 0.1


def test_lbm_forward_with_lpips_pixel_loss(cuda):
    """pixel_loss_type="lpips": LBMModel owns an LPIPS module (state-dict keys lpips_loss.*), the loss is finite and
    gradients reach the denoiser through LPIPS and the frozen VAE decoder."""
    from lbm_amd.models.embedders import ConditionerWrapper
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    from lbm_amd.models.schedulers import FlowMatchEulerDiscreteScheduler
    from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
    from lbm_amd.models.vae import AutoencoderKL, AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig
    from test_backward_gpu import SMALL_UNET
    torch.manual_seed(0)
    den = DiffusersUNet2DCondWrapper(**SMALL_UNET).to(BF)
    core = AutoencoderKL(scaling_factor=0.13025).to(BF)
    vae = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version="unused", tiling_size=(128, 128)), vae_model=core)
    cfg = LBMConfig(source_key="source", target_key="target", latent_loss_type="l2", latent_loss_weight=1.0,
                    pixel_loss_type="lpips", pixel_loss_weight=10.0, pixel_loss_max_size=64,
                    timestep_sampling="custom_timesteps", selected_timesteps=[250, 500, 750, 1000],
                    prob=[0.25, 0.25, 0.25, 0.25], bridge_noise_sigma=0.005)
    model = LBMModel(cfg, denoiser=den, training_noise_scheduler=FlowMatchEulerDiscreteScheduler(),
                     sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler(), vae=vae,
                     conditioner=ConditionerWrapper([]))
    _randomise(model.lpips_loss)
    model = model.to(BF).to(cuda)
    model.vae.freeze()
    assert "lpips_loss.net.slice3.12.weight" in model.state_dict() and "lpips_loss.lins.4.model.1.weight" in model.state_dict()
    model.train()
    g = torch.Generator().manual_seed(5)
    imgs = {k: (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).to(BF).to(cuda) for k in ("target", "source")}
    out = model(imgs, timesteps=torch.tensor([500, 250], device=cuda))
    assert float(out["pixel_recon_loss"].detach()) > 0 and torch.isfinite(out["loss"].detach())
    out["loss"].backward()
    grads = [p.grad for p in model.denoiser.parameters()]
    assert all(gd is not None and torch.isfinite(gd.float()).all() for gd in grads)
    assert all(p.grad is None for n, p in model.lpips_loss.named_parameters() if n.startswith("net."))


def test_refine_unet_gradients_and_lpips_training_step(cuda):
    """the pixel-space refine denoiser (UNet2DModel, vae=None): parameter gradients against autograd through the fp32
    restatement, then one trainer step of the refine model with the LPIPS pixel loss (the shipped refine configs)."""
    from lbm_amd.models.unets import DiffusersUNet2DWrapper
    from lbm_amd.inference.relight import build_filllight_refine_model
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    from oracle import diffusers_ref as dr
    from test_models_gpu import SMALL_UNET2D
    torch.manual_seed(3)
    m = DiffusersUNet2DWrapper(**SMALL_UNET2D).to(BF)
    sd = {k: v.detach().float().clone().requires_grad_() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    x, rgb, dy = (torch.randn(2, 3, 64, 64, generator=g).to(BF) for _ in range(3))
    ts = torch.tensor([1000.0, 500.0])
    ref = dr.unet2d_forward(sd, SMALL_UNET2D, torch.cat([x, rgb], 1).float(), ts)
    ref.backward(dy.float())
    m = m.to(cuda).train()
    out = m(sample=x.to(cuda), timestep=ts.to(cuda), conditioning={"cond": {"concat": rgb.to(cuda)}})
    assert rel_l2(out, ref) < 1e-2
    out.backward(dy.to(cuda))
    num = den = 0.0
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        a, b = p.grad.detach().float().cpu(), sd[name].grad
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
    assert (num / den) ** 0.5 < 6e-2

    model = build_filllight_refine_model(backbone_signature="offline", allow_random_init=True, block_out_channels=[64, 64, 128], pixel_loss_type="lpips",
                                         pixel_loss_weight=1.0, bridge_noise_sigma=0.005)
    _randomise(model.lpips_loss)
    model = model.to(BF).to(cuda)
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=1e-4, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
    pipe.configure_optimizers()
    assert all(not p.requires_grad for p in model.lpips_loss.parameters())
    model.train()
    batch = {"source": (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).to(BF).to(cuda),
             "target": (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).to(BF).to(cuda)}
    batch["rgb"] = batch["source"]
    losses = [float(pipe.fit_step(batch)["loss"]) for _ in range(3)]
    assert all(l == l and l > 0 for l in losses)
    assert pipe.global_step == 3
