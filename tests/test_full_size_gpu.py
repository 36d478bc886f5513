"""The BASELINE configurations at their real sizes on the GPU, through properties that need no full-size CPU oracle
pass: 2048x2048 inference (256x256 latent: 9 decode tiles per image with ragged 32-wide last tiles, 65,536-token
encoder attention, 16,384-token UNet attention), the batch-16 1024x1024 headline step on two streams, and the
512x512 batch-8 training step of the 2.43 B-parameter model."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def full_model(cuda):
    import bench
    return bench.build_model(cuda, 1024)


def test_c5_2048_inference_true_geometry(full_model, cuda):
    """BASELINE config 5 (2048x2048, 1 NFE) at its true geometry, batch 2: finite; an image returns the same bits
    alone and in the batch; and the device tile merge (src/lbm/models/vae/autoencoderKL.py:88-131,
    src/lbm/models/utils.py:13-103, 206-258: tile starts 0 / 112 / 224, last tiles 32 wide and zero-padded, the
    asymmetric gaussian weights at 1024 / 256 pixels) equals the restated reference tiling driven with the HIP
    UNTILED decode of each padded 128x128 tile."""
    import bench
    from oracle import bridge_ref as br
    model = full_model
    g = torch.Generator().manual_seed(77)
    imgs = (torch.rand(2, 3, 2048, 2048, generator=g) * 2 - 1).to(BF).to(cuda)
    eps = torch.randn(2, 4, 256, 256, generator=g).to(BF).to(cuda)
    out2 = bench.one_step(model, imgs, eps, 1)
    out1 = bench.one_step(model, imgs[1:2].contiguous(), eps[1:2].contiguous(), 1)
    assert out2.shape == (2, 3, 2048, 2048) and out2.dtype == torch.float32       # the tiled decode returns f32, like the reference's CPU merge
    assert bool(torch.isfinite(out2).all())
    assert torch.equal(out2[1:2], out1)
    assert float((out2[0] - out2[1]).abs().mean()) > 1e-3                          # two different images
    # the sampled latent, then the reference's tiling around the HIP decoder
    with torch.no_grad():
        z = model.vae.encode(imgs, eps=eps)
        vae, model.vae = model.vae, None
        try:
            lat = model.sample(z, num_steps=1)
        finally:
            model.vae = vae
        assert lat.shape == (2, 4, 256, 256)
        merged = model.vae.decode(lat)
        assert torch.equal(merged, out2)

        def decode_tile(t):           # t: f32 CPU [2, 4, 128, 128], zero-padded by the oracle's tiler
            return model.vae.decode(t.to(BF).to(cuda)).float().cpu()

        # the wrapper un-scales the whole latent once and tiles after; un-scaling tile by tile gives the same bf16 values
        ref = br.tiled_decode(lat.float().cpu(), decode_tile, tile=(128, 128), overlap=(16, 16), scale=8)
    assert ref.shape == (2, 3, 2048, 2048)
    err = rel_l2(merged, ref)
    worst = float((merged.cpu() - ref).abs().max())
    print(f"2048x2048 tiled decode, device merge vs restated reference tiling: rel-L2 {err:.2e}, max abs {worst:.2e}")
    assert err < 2e-6 and worst < 1e-4


def _pair(make, cuda, scale_qk):
    """the same layer twice: bf16 (the fused kernels) and held in f32 (fp32 mode: exact-f32 MFMA products, scores
    materialised, library expf), with to_q / to_k scaled up so that the softmax is far from uniform"""
    torch.manual_seed(3)
    lo = make().to(BF)
    with torch.no_grad():
        lo.to_q.weight.mul_(scale_qk)
        lo.to_k.weight.mul_(scale_qk)
        # ... and the value / output projections, so that the attention branch is as large as the residual it is added
        # to (the bf16 layer rounds the SUM: a small branch would be measured through the rounding of the residual)
        lo.to_v.weight.mul_(4.0)
        lo.to_out[0].weight.mul_(4.0)
    hi = make()
    hi.load_state_dict({k: v.float() for k, v in lo.state_dict().items()})
    return lo.to(cuda).eval(), hi.to(cuda).eval()


def test_c5_attention_layers_at_true_token_counts(cuda):
    """The two attention shapes only the 2048x2048 configuration reaches: the VAE encoder's mid-block attention over
    65,536 tokens (one head of 512: the two-pass bf16-probability path, one image per chunk) and the UNet's
    level-1 self-attention over 16,384 tokens (10 heads of 64: the flash kernel), each against the same layer in
    fp32 mode (itself <= 1e-5 from the CPU oracle, tests/test_fp32_gpu.py)."""
    from lbm_amd.models.unets.unet2d_condition import Attention
    from lbm_amd.models.vae.autoencoder_kl import VaeAttention
    g = torch.Generator().manual_seed(9)
    # VAE mid attention, N = 256 x 256
    lo, hi = _pair(lambda: VaeAttention(512), cuda, 3.0)
    x = torch.randn(1, 256, 256, 512, generator=g).to(BF).to(cuda)
    with torch.no_grad():
        y_lo = lo(x)
        y_hi = hi(x.float())
    torch.cuda.synchronize()
    branch_lo, branch_hi = y_lo.float() - x.float(), y_hi - x.float()
    e_vae = rel_l2(branch_lo, branch_hi)
    print(f"VAE mid attention, 65,536 tokens: attention branch rel-L2 {e_vae:.4f} (bf16 two-pass vs fp32 mode)")
    assert bool(torch.isfinite(y_lo.float()).all()) and float(branch_hi.abs().mean()) > 0.3
    assert e_vae < 1e-2
    del lo, hi, x, y_lo, y_hi, branch_lo, branch_hi
    torch.cuda.empty_cache()
    # UNet level-1 self-attention, N = 128 x 128, 10 heads of 64
    lo, hi = _pair(lambda: Attention(640, 10, 64), cuda, 3.0)
    xn = torch.randn(2, 16384, 640, generator=g).to(BF).to(cuda)
    zero_lo = torch.zeros_like(xn)
    with torch.no_grad():
        o_lo = lo(xn, None, zero_lo)
        o_hi = hi(xn.float(), None, zero_lo.float())
    torch.cuda.synchronize()
    e_unet = rel_l2(o_lo, o_hi)
    print(f"UNet self-attention, 16,384 tokens x 10 heads of 64: rel-L2 {e_unet:.4f} (bf16 flash vs fp32 mode)")
    assert bool(torch.isfinite(o_lo.float()).all())
    assert e_unet < 1e-2


def test_headline_batch16_two_streams_full_size(full_model, cuda):
    """The headline step itself (1024x1024, batch 16, the 2.43 B-parameter model): the two-stream split returns the
    same bits twice in a row, the same bits as one stream, and images 3..5 run as a batch of three are members 3..5."""
    import bench
    from lbm_amd import streams
    model = full_model
    g = torch.Generator().manual_seed(16)
    images = (torch.rand(16, 3, 1024, 1024, generator=g) * 2 - 1).to(BF).to(cuda)
    eps = torch.randn(16, 4, 128, 128, generator=g).to(BF).to(cuda)
    assert streams._SPLIT
    a = bench.one_step(model, images, eps, 1)
    b = bench.one_step(model, images, eps, 1)
    old, streams._SPLIT = streams._SPLIT, False
    try:
        c = bench.one_step(model, images, eps, 1)
    finally:
        streams._SPLIT = old
    d = bench.one_step(model, images[3:6].contiguous(), eps[3:6].contiguous(), 1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a.float()).all())
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(d, a[3:6])


def test_full_size_training_step(cuda):
    """BASELINE config 4 at full size (512x512, batch 8, 2.43 B-parameter UNet, two frozen VAE encodes, latent loss,
    AdamW): the loss is finite and falls over three steps on a fixed batch, and a second run from the same seeds
    reproduces every loss and the updated parameters bit for bit."""
    import bench
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    g = torch.Generator().manual_seed(4242)
    batch = {"source_image": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(cuda),
             "source_image_paste": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(cuda)}
    ts = torch.tensor([250, 500, 750, 1000, 250, 500, 750, 1000], device=cuda)
    bridge_eps = torch.randn(8, 4, 64, 64, generator=g).to(BF).to(cuda)

    def run():
        model = bench.build_model(cuda, 512)
        model.train()
        model.vae.freeze()
        forward = model.forward
        model.forward = lambda b, *a, **k: forward(b, bridge_eps=bridge_eps, timesteps=ts)
        pipe = TrainingPipeline(model, TrainingConfig(learning_rate=1e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
        pipe.configure_optimizers()
        losses = []
        for _ in range(3):
            torch.manual_seed(31)            # the frozen VAE's posterior draws: the same latents every step
            losses.append(float(pipe.fit_step(batch)["loss"]))
        torch.cuda.synchronize()
        flat = pipe.optims[0].groups[0].flat
        digest = (float(flat.float().sum()), float(flat.float().abs().sum()), flat[:: 1 << 16].clone().cpu())
        del pipe, model
        torch.cuda.empty_cache()
        return losses, digest

    l1, d1 = run()
    l2, d2 = run()
    print(f"full-size training step, losses over three steps: {l1}")
    assert all(math.isfinite(v) for v in l1)
    assert l1[2] < l1[0]
    assert l1 == l2
    assert d1[0] == d2[0] and d1[1] == d2[1] and torch.equal(d1[2], d2[2])
