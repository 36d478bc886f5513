#!/usr/bin/env python
"""Training step of the fill-light model as the reference ships it (examples/training/config/fill_lighting_simple.yaml,
train_fill_lighting_simple.py): SD1.5-style UNet (865 M parameters, 5 input channels = latent + depth mask), five
lighting tokens from the LightingParamsEmbedder into a 768-wide cross-attention, latent l2 + pixel l2 loss through the
frozen VAE decoder, denoiser and conditioner trained (conditioner at 10x the learning rate), AdamW, 512^2, batch 2.

  python scripts/bench_train_filllight.py --steps 3 --warmup 1 [--batch 2] [--pixel-loss l2|lpips|none]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import dist as D

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--pixel-loss", default="l2")
    args = ap.parse_args()
    rank, local_rank, world = D.env_world()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = D.init("nccl", device)
    from lbm_amd.inference.relight import build_filllight_model
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    torch.manual_seed(0)
    model = build_filllight_model(
        backbone_signature="offline", allow_random_init=True, unet_input_channels=5, timestep_sampling="custom_timesteps",
        selected_timesteps=[250, 500, 750, 1000], prob=[0.25, 0.25, 0.25, 0.25],
        pixel_loss_type=args.pixel_loss if args.pixel_loss != "none" else "l2",
        pixel_loss_weight=0.0 if args.pixel_loss == "none" else 1.0, latent_loss_type="l2", latent_loss_weight=1.0,
        bridge_noise_sigma=0.005, conditioning_images_keys=[], conditioning_masks_keys=["depth"],
        lighting_embedder_config=dict(mapping_size=256, scale=10.0, hidden_dim=512, out_dim=768, seq_len=5,
                                      param_groups=[[0, 1], [2], [3, 4, 5], [6], [7]]),
        lighting_condition_weight=10.0, concat_condition_weight=1.0)
    model = model.to(BF).to(device)
    model.train()
    model.vae.freeze()
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW",
                                                  trainable_params=["denoiser.*", "conditioner.*"]), dist=dist)
    pipe.configure_optimizers()
    g = torch.Generator().manual_seed(1234 + rank)
    B, R = args.batch, args.res
    batch = {"source": (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(BF).to(device),
             "target": (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(BF).to(device),
             "depth": torch.rand(B, 1, R, R, generator=g).to(BF).to(device),
             "lighting_params": torch.rand(B, 8, generator=g).to(device)}
    for _ in range(args.warmup):
        loss = pipe.fit_step(batch)["loss"]
    torch.cuda.synchronize(); D.barrier(dist); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = pipe.fit_step(batch)["loss"]
    torch.cuda.synchronize(); D.barrier(dist); torch.cuda.synchronize()
    el = D.max_over_ranks(time.perf_counter() - t0, dist, device)
    assert bool(torch.isfinite(loss))
    if rank == 0:
        sps = B * world * args.steps / el
        print(json.dumps({"metric": f"training samples/sec ({R}^2, fill-light SD1.5-style UNet + lighting tokens, latent l2 + pixel {args.pixel_loss}, AdamW)",
                          "value": round(sps, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "batch_per_gpu": B, "ms_per_step": round(el / args.steps * 1e3, 1), "dtype": "bf16", "data": "synthetic",
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1), "loss": float(loss),
                          "trainable_params": sum(p.numel() for g_ in pipe.optims[0].groups for p in g_.params)}))


if __name__ == "__main__":
    main()
