#!/usr/bin/env python
"""Headline benchmark: images/sec of 1024x1024 1-NFE LBM inference (VAE encode -> one
SDXL-style UNet evaluation + Euler step -> VAE decode), bf16, batch 16 per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus 8 --steps 5 --warmup 2        (starts its own 8 ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; inference shards by image batch, so ranks never exchange data (the
only collectives are the timing barrier and the max-over-ranks of the elapsed time).
Rank 0 prints ONE JSON line.  Inputs are synthetic (uniform [-1, 1] images, seeded random
weights of the reference's architecture: no checkpoint is available offline) and resident
in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF = torch.bfloat16
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0                    # MI355X HBM3E


def build_model(device, res):
    from lbm_amd.inference.utils import sdxl_style_unet_kwargs
    from lbm_amd.models.embedders import ConditionerWrapper
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    from lbm_amd.models.schedulers import FlowMatchEulerDiscreteScheduler
    from lbm_amd.models.unets import DiffusersUNet2DCondWrapper
    from lbm_amd.models.vae import AutoencoderKL, AutoencoderKLDiffusers, AutoencoderKLDiffusersConfig

    torch.manual_seed(0)
    with torch.device(device):      # parameters are created and initialised on the GPU
        unet = DiffusersUNet2DCondWrapper(**sdxl_style_unet_kwargs()).to(BF)
        core = AutoencoderKL(scaling_factor=0.13025).to(BF)
    vae = AutoencoderKLDiffusers(AutoencoderKLDiffusersConfig(version="synthetic", tiling_size=(128, 128)), vae_model=core)
    cfg = LBMConfig(source_key="source_image", target_key="source_image_paste", bridge_noise_sigma=0.005,
                    timestep_sampling="custom_timesteps", selected_timesteps=[250.0, 500.0, 750.0, 1000.0],
                    prob=[0.25, 0.25, 0.25, 0.25])
    model = LBMModel(cfg, denoiser=unet, sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler(),
                     training_noise_scheduler=FlowMatchEulerDiscreteScheduler(), vae=vae,
                     conditioner=ConditionerWrapper([]))
    model.to(device).to(BF)
    model.eval()
    return model


@torch.no_grad()
def one_step(model, images, enc_eps, nfe):
    """One pass of the hot path over one batch: encode -> n-step bridge sampling -> decode."""
    z = model.vae.encode(images, eps=enc_eps)
    return model.sample(z, num_steps=nfe, conditioner_inputs={"source_image": images})


def cpu_baseline(model, threads):
    """The fp32 CPU oracle (a restatement of the reference's diffusers-based path, kind 'port') on a
    bounded sample: 512x512 images, 1 NFE (the reference's CPU-runnable configuration); one untimed warm-up
    pass, then the median of three timed passes."""
    from oracle import bridge_ref as br
    from oracle import diffusers_ref as dr
    torch.set_num_threads(threads)
    usd = {k: v.detach().float().cpu() for k, v in model.denoiser.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in model.vae.vae_model.state_dict().items()}
    from lbm_amd.inference.utils import sdxl_style_unet_kwargs
    cfg = sdxl_style_unet_kwargs()
    n = 2
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(n, 3, 512, 512, generator=g) * 2 - 1
    eps = torch.randn(n, 4, 64, 64, generator=torch.Generator().manual_seed(4321))

    def one_pass():
        t0 = time.perf_counter()
        with torch.no_grad():
            z = dr.posterior_sample(dr.vae_encode_moments(vsd, x), eps) * 0.13025
            lat = br.sample(z, lambda s, t: dr.unet_forward(usd, cfg, s, t), 1, [], 0.005)
            img = dr.vae_decode(vsd, lat / 0.13025)
        dt = time.perf_counter() - t0
        assert torch.isfinite(img).all()
        return dt

    one_pass()
    times = sorted(one_pass() for _ in range(3))
    dt = times[1]
    return {"value": round(n / dt, 4), "unit": "images/s (512x512)", "cores": threads, "kind": "port",
            "sample": f"{n} images 512x512 per pass, 1 NFE, fp32 torch CPU oracle; 1 warm-up pass, median of 3 timed "
                      f"passes ({times[0]:.1f} / {times[1]:.1f} / {times[2]:.1f} s) "
                      f"(5.33 TFLOP/image; the 1024x1024 workload is 23.25 TFLOP/image, i.e. "
                      f"~{n / dt * 5.33 / 23.25:.3f} images/s extrapolated by FLOPs)"}


def measured_gemm_ceiling(device, reps=5):
    """What the matrix pipe delivers on this box under a large dense product of random bf16 data (16384 x 16384 x 8192,
    one igemm launch per rep, HIP events on the launch stream): the DVFS-limited ceiling next to the nominal peak."""
    from lbm_amd import ops
    g = torch.Generator(device=device).manual_seed(5)
    a = torch.randn(16384, 8192, generator=g, device=device, dtype=torch.float32).to(BF)
    b = torch.randn(16384, 8192, generator=g, device=device, dtype=torch.float32).to(BF)
    out = torch.empty(16384, 16384, dtype=BF, device=device)
    ops.gemm_nt(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_nt(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"value": round(2.0 * 16384 * 16384 * 8192 / (ms * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
            "shape": "16384 x 16384 x 8192 bf16 NT product, random N(0,1) operands, lbmk_igemm", "ms": round(ms, 3), "reps": reps}


PMC_TRAFFIC_FILES = ("profiles/r03_pmc_igemm_step_traffic.json", "profiles/r02_pmc_igemm_step_traffic_v2.json",
                     "profiles/r02_pmc_igemm_step_traffic.json", "profiles/r01_pmc_igemm_step_traffic.json")


def pmc_traffic(args, launches):
    """HBM bytes per igemm launch, REPLAYED from the newest committed record of separate rocprofv3 --pmc passes over this
    same command (hardware counters cannot be read from inside the run): the record's bytes per STEP over the ``launches``
    of the one-stream pass the roofline is measured on (the two-stream step of the PMC passes cuts the same bytes into
    about twice as many launches).  (bytes, source file), or (None, None) when the workload differs from the profiled one
    or no record exists."""
    for rel in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                rec = json.load(f)
        except OSError:
            continue
        if rec.get("workload") == {"res": args.res, "batch": args.batch, "nfe": args.nfe}:
            if "traffic_bytes_per_step" in rec:
                return round(rec["traffic_bytes_per_step"] / launches), rel
            return round(rec["traffic_bytes_per_launch"]), rel
    return None, None


# algorithmic forward FLOPs per image (FLOP = 2 * MAC), SDXL-style UNet + SD VAE: resolution -> (encode, unet, decode)
FLOPS_T = {512: (1.117, 1.693, 2.515), 1024: (4.879, 7.898, 10.470), 2048: (26.113, 49.630, 94.23)}


def _timed_steps(fn, steps, dist, device):
    from lbm_amd import dist as D
    torch.cuda.synchronize()
    D.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    D.barrier(dist)
    torch.cuda.synchronize()
    return D.max_over_ranks(time.perf_counter() - t0, dist, device)


def extra_inference(model, device, rank, world, dist, res, batch, nfe, steps):
    """one more inference configuration (per-GPU share of a BASELINE config), measured after the headline region"""
    g = torch.Generator().manual_seed(99 + rank)
    images = (torch.rand(batch, 3, res, res, generator=g) * 2 - 1).to(BF).to(device)
    eps = torch.randn(batch, 4, res // 8, res // 8, generator=g).to(BF).to(device)
    out = one_step(model, images, eps, nfe)                      # warm-up at this shape
    el = _timed_steps(lambda: one_step(model, images, eps, nfe), steps, dist, device)
    assert bool(torch.isfinite(out.float()).all())
    ips = batch * world * steps / el
    enc, unet, dec = FLOPS_T[res]
    tf = ips * (enc + nfe * unet + dec)
    del images, eps, out
    return {"workload": f"{res}x{res}, {nfe}-NFE, batch {batch}/GPU x{world}" + (", tiled decode (9 tiles/image)" if res > 1024 else ""),
            "value": round(ips, 3), "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 1), "steps": steps,
            "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / world / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}


def extra_fp32(device, rank, world, dist, res, batch, steps):
    """BASELINE config 1's dtype: the same model held in f32 (fp32 mode, exact-f32 MFMA), 1 NFE, batch 1 per GPU"""
    from lbm_amd import ops
    model = build_model(device, res).to(torch.float32)
    g = torch.Generator().manual_seed(7 + rank)
    images = (torch.rand(batch, 3, res, res, generator=g) * 2 - 1).to(device)
    eps = torch.randn(batch, 4, res // 8, res // 8, generator=g).to(device)
    out = one_step(model, images, eps, 1)
    el = _timed_steps(lambda: one_step(model, images, eps, 1), steps, dist, device)
    assert out.dtype == torch.float32 and bool(torch.isfinite(out).all())
    ips = batch * world * steps / el
    tf = ips * sum(FLOPS_T[res])
    del model
    torch.cuda.empty_cache()
    return {"workload": f"{res}x{res}, 1-NFE, batch {batch}/GPU x{world}, fp32 mode (f32 activations, v_mfma_f32_32x32x2_f32)",
            "value": round(ips, 3), "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 1), "steps": steps,
            "tflops": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / world / 157.3, 4)}


def extra_training(model, device, rank, world, dist, batch, steps, resolutions=(512, 1024)):
    """bridge-matching training step: two frozen VAE encodes, interpolant, UNet forward + backward, latent loss,
    gradient exchange over RCCL when world > 1, fused AdamW.  7.31 TFLOP per 512x512 sample (2 * 1.117 + 3 * 1.693).
    One pipeline, measured at each resolution: 512x512 is the size the reference's training configs use
    (examples/training/config/*.yaml: image_size 512); 1024x1024 shows the same kernels on four times the rows per
    GEMM (the 512x512 step's token matrices have 2048 / 8192 rows)."""
    from lbm_amd.trainer import TrainingConfig, TrainingPipeline
    model.train()
    model.vae.freeze()
    pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]),
                            dist=dist)
    pipe.configure_optimizers()
    out = {}
    for res in resolutions:
        g = torch.Generator().manual_seed(4242 + rank)
        batch_d = {"source_image": (torch.rand(batch, 3, res, res, generator=g) * 2 - 1).to(BF).to(device),
                   "source_image_paste": (torch.rand(batch, 3, res, res, generator=g) * 2 - 1).to(BF).to(device)}
        torch.cuda.reset_peak_memory_stats()
        for _ in range(2):
            loss = pipe.fit_step(batch_d)["loss"]
        el = _timed_steps(lambda: pipe.fit_step(batch_d), steps, dist, device)
        assert bool(torch.isfinite(loss))
        sps = batch * world * steps / el
        enc, unet, _ = FLOPS_T[res]
        tf = sps * (2 * enc + 3 * unet)
        out[f"train_{res}_b{batch}"] = {
            "workload": f"training step {res}x{res}, batch {batch}/GPU x{world}: 2 frozen VAE encodes + UNet fwd/bwd + latent loss + "
                        f"{pipe.reducer.mode if dist is not None else 'no'} gradient exchange + AdamW",
            "value": round(sps, 3), "unit": "samples/s", "ms_per_step": round(el / steps * 1e3, 1), "steps": steps,
            "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / world / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        del batch_d
        torch.cuda.empty_cache()
    return out


def launch_ranks(n, argv, timeout_s=None):
    """Starts ``n`` ranks of this script (one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in each child's
    environment, rendezvous on 127.0.0.1 at a free port), as the reference's entry point starts its own ranks
    (examples/training/train_base.py:114-132: devices=torch.cuda.device_count()).  The children inherit stdout, so
    rank 0's JSON line is this command's output.  Returns 0 when every rank did, else the first non-zero status;
    when one rank fails the others are terminated (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    status, t0 = 0, time.time()
    live = list(procs)
    while live:
        for pr in list(live):
            rc = pr.poll()
            if rc is None:
                continue
            live.remove(pr)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                print(f"bench.py: rank {procs.index(pr)} exited with status {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                for other in live:
                    other.terminate()
        if live and timeout_s is not None and time.time() - t0 > timeout_s:
            status = status or 124
            for other in live:
                other.kill()
        time.sleep(0.05)
    return status


def dry_run(args, rank, world):
    """--dry-run: the rank plumbing of the benchmark without a GPU (gloo): rendezvous, barrier, max over ranks, and
    rank 0's one JSON line.  LBM_BENCH_DRY_FAIL_RANK=r makes rank r exit with status 7 before the rendezvous (the
    launcher must then stop the others and report failure)."""
    from lbm_amd import dist as D
    if os.environ.get("LBM_BENCH_DRY_FAIL_RANK") == str(rank):
        return 7
    dist = D.init("gloo")
    t0 = time.perf_counter()
    D.barrier(dist)
    el = D.max_over_ranks(time.perf_counter() - t0 + rank, dist)      # "+ rank": the max must come from the last rank
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks_s": round(el, 3)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--nfe", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-records of the other BASELINE configs")
    ap.add_argument("--extras-budget-s", type=float, default=420.0,
                    help="wall-clock limit for everything after the headline region; on expiry the line is printed without "
                         "what is missing and the process exits")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank plumbing only, on the CPU over gloo: rendezvous, barrier, max over ranks, one JSON line "
                         "(what the launcher test runs where there is no GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (and never will); it starts the N
        # ranks as child processes, relays what they print and exits with the first failing status
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from lbm_amd import dist as D
    rank, local_rank, world = D.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = D.init("nccl", device)       # nccl = RCCL; used for the timing barrier / max only

    from lbm_amd import ops
    model = build_model(device, args.res)
    g = torch.Generator().manual_seed(1234 + rank)
    images = (torch.rand(args.batch, 3, args.res, args.res, generator=g) * 2 - 1).to(BF).to(device)
    L = args.res // 8
    enc_eps = torch.randn(args.batch, 4, L, L, generator=torch.Generator().manual_seed(4321 + rank)).to(BF).to(device)

    # set-up, not a benchmark step: the first pass packs every layer's weights into the kernels' layout and grows
    # torch's caching allocator to the working set (reported as "setup_passes")
    out = one_step(model, images, enc_eps, args.nfe)
    for _ in range(args.warmup):
        out = one_step(model, images, enc_eps, args.nfe)
    torch.cuda.synchronize()
    D.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step(model, images, enc_eps, args.nfe)
    torch.cuda.synchronize()
    D.barrier(dist)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert out.shape == (args.batch, 3, args.res, args.res) and bool(torch.isfinite(out.float()).all())
    elapsed = D.max_over_ranks(elapsed, dist, device)

    import threading
    state = {"roofline": None, "secondary": None, "cpu": None, "extra": {}, "done": False, "in_flight": "roofline pass"}
    emit_lock = threading.Lock()

    def emit():
        with emit_lock:
            if state["done"]:
                return
            state["done"] = True
        if rank != 0:
            return
        total_images = args.batch * world * args.steps
        line = {
            "metric": f"images/sec ({args.res}^2, {args.nfe}-NFE LBM) whole node", "value": round(total_images / elapsed, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_passes": 1,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "ms_per_image": round(elapsed / total_images * 1e3 * world, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (uniform [-1,1] images, seeded random-init weights of the reference architecture)",
            "config": {"workload": f"LBM inference {args.res}x{args.res}, {args.nfe}-NFE, SDXL-style UNet (2.43B) + SD VAE encode/decode, "
                                   f"batch {args.batch}/GPU, data-parallel x{world}", "global_batch": args.batch * world,
                       "flops_per_image_T": 23.25 if (args.res, args.nfe) == (1024, 1) else None},
            "roofline": state["roofline"], "secondary_kernels": state["secondary"], "cpu_baseline": state["cpu"],
            "extra": state["extra"] or None,
        }
        print(json.dumps(line), flush=True)

    # everything below is measured AFTER the headline region; a watchdog bounds it so that a stall there (first
    # multi-rank collective of the training record, a slow host) can never cost the headline line;
    # costs the process its clean exit status but not the headline line: exit status 3 on every rank
    def expire():
        state["extra"]["aborted"] = (f"post-headline measurements exceeded {args.extras_budget_s:.0f} s while running: "
                                     f"{state['in_flight']}")
        emit()
        sys.stdout.flush()
        os._exit(3)

    watchdog = threading.Timer(args.extras_budget_s, expire)
    watchdog.daemon = True
    watchdog.start()

    roofline, cpu, secondary = None, None, None
    if rank == 0 and not args.no_roofline:
        # dominant kernel = the implicit-GEMM conv / linear kernel: every launch of one extra, untimed
        # step is bracketed with HIP events on the launch stream; algorithmic FLOPs = 2*M*N*K per launch of the
        # reference operation (an upsampling convolution counts its 9 taps although the four parity convolutions
        # that compute it multiply 4/9 of them: "executed" is the rate of the products actually multiplied)
        # This pass runs on ONE stream so that a launch's duration is its own: the timed steps above run the two halves
        # of the batch on two streams (lbm_amd.streams), where concurrent kernels stretch each other's durations.
        from lbm_amd import streams
        split, streams._SPLIT = streams._SPLIT, False
        try:
            ops.igemm_profile_begin()
            one_step(model, images, enc_eps, args.nfe)
            prof = ops.igemm_profile_end()
            # the same one-stream step without the per-launch events: what the kernels' shares are shares OF
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            one_step(model, images, enc_eps, args.nfe)
            e1.record()
            torch.cuda.synchronize()
            one_stream_ms = e0.elapsed_time(e1)
        finally:
            streams._SPLIT = split
        ceiling = measured_gemm_ceiling(device)
        tf = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "igemm_kernel (implicit-GEMM conv/linear, all tile shapes)",
                    "achieved": round(tf, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                    "executed": round(prof["executed"] / (prof["ms"] * 1e-3) / 1e12, 2), "traffic": pmc_traffic(args, prof["launches"])[0],
                    "traffic_source": (f"replayed constant from {pmc_traffic(args, prof['launches'])[1]} (separate rocprofv3 --pmc "
                                       "passes over this command; not measured in this run)") if pmc_traffic(args, prof["launches"])[1] else None,
                    "launches": prof["launches"], "avg_launch_ms": round(prof["ms"] / prof["launches"], 4),
                    "flops_per_launch": prof["flops"] / prof["launches"],
                    "measured_ceiling": ceiling,
                    "frac_of_measured_ceiling": round(tf / ceiling["value"], 4),
                    "one_stream_step_ms": round(one_stream_ms, 2),
                    "share_of_step_time": round(prof["ms"] / one_stream_ms, 3),
                    "measured_on": "one stream, one extra untimed step" + (
                        " (the timed steps overlap two half-batches on two streams)" if split else "")}
        # the other kernels of the step, same event-bracketed measurement: the attention kernel against the MFMA
        # peak, the normalisation / softmax passes against HBM (algorithmic bytes: 2 B read + 2 B written per element,
        # plus 2 B for a separate statistics read; softmax 4 B read + 2 B written)
        secondary = []
        step_ms = one_stream_ms
        for kind, (ms, work, n) in sorted(prof["aux"].items()):
            if kind == "flash_attn":
                ach, peak, unit, bound = work / (ms * 1e-3) / 1e12, MFMA_BF16_DENSE_PEAK_TFLOPS, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
            secondary.append({"kernel": kind, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                              "frac": round(ach / peak, 4), "launches": n, "share_of_step_time": round(ms / step_ms, 3)})
    state["roofline"], state["secondary"] = roofline, secondary
    if not args.no_extras and (args.res, args.nfe) == (1024, 1):
        # per-GPU shares of the other BASELINE configurations, every rank taking part (same barrier / max-over-ranks
        # timing as the headline): C1 = 512^2 batch 1 in fp32, C3 = 4-step sampling at 1024^2, C5 = 2048^2, C4 = the
        # training step
        for name, fn in (("c1_512_b1_fp32", lambda: extra_fp32(device, rank, world, dist, 512, 1, 2)),
                         ("nfe4_1024_b8", lambda: extra_inference(model, device, rank, world, dist, 1024, 8, 4, 2)),
                         ("res2048_b8", lambda: extra_inference(model, device, rank, world, dist, 2048, 8, 1, 1)),
                         ("train_b8", lambda: extra_training(model, device, rank, world, dist, 8, 3))):
            state["in_flight"] = name
            try:
                rec = fn()
                if name == "train_b8":
                    state["extra"].update(rec)          # train_512_b8, train_1024_b8
                else:
                    state["extra"][name] = rec
            except Exception as err:          # an extra must never cost the headline line
                state["extra"][name] = {"error": f"{type(err).__name__}: {err}"[:300]}
                break
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the box's CPU share for a one-GPU job is 16 cores (os.cpu_count() reports the whole machine)
        state["in_flight"] = "cpu_baseline"
        state["cpu"] = cpu_baseline(model, min(16, os.cpu_count() or 1))
    watchdog.cancel()
    emit()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
