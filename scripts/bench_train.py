#!/usr/bin/env python
"""Training-step benchmark: 512x512 images, batch 8 per GPU, bf16, SDXL-style UNet:
two frozen VAE encodes (forward only) + bridge interpolant + UNet forward/backward + latent loss + gradient
all-reduce (RCCL, N > 1) + fused AdamW.  7.31 TFLOP per sample (2*1.117 + 3*1.693).

  python scripts/bench_train.py --steps 3 --warmup 1
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train.py --gpus N
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd import dist as D

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--ops", action="store_true", help="after timing, count aten ops of one step by calling line")
    args = ap.parse_args()
    rank, local_rank, world = D.env_world()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = D.init("nccl", device)
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    model = bench.build_model(device, args.res)
    model.train()
    model.vae.freeze()
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]),
                            dist=dist)
    pipe.configure_optimizers()
    g = torch.Generator().manual_seed(1234 + rank)
    batch = {"source_image": (torch.rand(args.batch, 3, args.res, args.res, generator=g) * 2 - 1).to(BF).to(device),
             "source_image_paste": (torch.rand(args.batch, 3, args.res, args.res, generator=g) * 2 - 1).to(BF).to(device)}
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
        sps = args.batch * world * args.steps / el
        print(json.dumps({"metric": "training samples/sec (512^2, latent loss, SDXL-style UNet fwd+bwd, AdamW)", "value": round(sps, 3),
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 1), "dtype": "bf16", "data": "synthetic",
                          "achieved_tflops": round(sps * 7.31, 1), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                          "loss": float(loss)}))


    if args.ops and rank == 0:
        from collections import Counter
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
            pipe.fit_step(batch)
            torch.cuda.synchronize()
        cnt = Counter()
        for ev in prof.events():
            if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::contiguous", "aten::cat", "aten::fill_", "aten::mul"):
                frames = [f for f in (ev.stack or []) if "lbm_amd" in f]
                cnt[(ev.name, frames[0].split("lbm_amd/")[-1] if frames else "autograd engine")] += 1
        for (name, where), n in cnt.most_common(40):
            print(f"{n:6d} {name:18s} {where}")


if __name__ == "__main__":
    main()
