"""world_size-2 test of the data-parallel plumbing on CPU (gloo): batch sharding covers every
image exactly once with no exchange, and the timing reduction is a max over ranks."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
from lbm_amd import dist as D
rank, local_rank, world = D.env_world()
d = D.init("gloo")
assert d is not None and d.get_world_size() == 2
lo, hi = D.shard_range(37, rank, world)
# every rank works on its own slice only; gather just to let the test check the partition
mine = torch.zeros(37, dtype=torch.int32); mine[lo:hi] = 1
d.all_reduce(mine)
assert bool((mine == 1).all()), mine
D.barrier(d)
t = D.max_over_ranks(1.0 + rank, d)
assert t == 2.0, t
if rank == 0:
    print(json.dumps({"ok": True, "slice0": [lo, hi]}))
d.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert '"ok": true' in outs[0][0]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it starts two ranks itself (as the reference's entry point
    starts one process per device, examples/training/train_base.py:114-132) and relays rank 0's JSON line; a rank that
    fails makes the whole command fail.  --dry-run: the rank plumbing only, on the CPU over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["max_over_ranks_s"] >= 1.0                       # rank 1 contributes its elapsed time + 1 s
    bad = subprocess.run(cmd, env=dict(env, LBM_BENCH_DRY_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode == 7, (bad.returncode, bad.stderr[-2000:])
    assert "rank 1 exited with status 7" in bad.stderr
    # under a launcher (WORLD_SIZE set) a mismatching --gpus is still refused
    worse = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert worse.returncode != 0 and "WORLD_SIZE=1" in worse.stderr


def test_shard_range_partitions():
    from lbm_amd.dist import shard_range
    for total in (0, 1, 16, 37, 64):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                cover += list(range(lo, hi))
            assert cover == list(range(total))
    assert shard_range(64, 3, 8) == (24, 32)


def _reducer_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import importlib.util, sys, types
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the reducer is device-agnostic; load it without the HIP library (ops is only used by the optimizer)
    pkg = types.ModuleType("lbm_amd"); pkg.__path__ = [os.path.join(os.path.dirname(os.path.dirname(__file__)), "lbm_amd")]
    sys.modules.setdefault("lbm_amd", pkg)
    sys.modules.setdefault("lbm_amd.ops", types.ModuleType("lbm_amd.ops"))
    sys.modules["lbm_amd"].ops = sys.modules["lbm_amd.ops"]
    from lbm_amd.trainer.trainer import FlatGroup, GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    net[3].weight.requires_grad_(True)
    unused = torch.nn.Parameter(torch.zeros(5))               # never receives a gradient
    params = list(net.parameters()) + [unused]
    grp = FlatGroup(params, lr=1e-3, bucket_bytes=64)          # several small buckets
    red = GradReducer([grp], dist, mode="all_reduce")
    x = torch.randn(6, 16, generator=torch.Generator().manual_seed(100 + rank))
    red.begin_step()
    net(x).pow(2).sum().backward()
    scale = red.finish()[0]
    flat_mean = grp.grad * scale
    # reference: gradients of both ranks' inputs computed locally
    ref = torch.zeros_like(grp.grad)
    for r in range(world):
        net2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
        net2.load_state_dict(net.state_dict())
        xr = torch.randn(6, 16, generator=torch.Generator().manual_seed(100 + r))
        net2(xr).pow(2).sum().backward()
        g2 = list(net2.parameters()) + [None]
        for p, o, q2 in zip(grp.params, grp.offsets, g2[::-1]):
            if q2 is not None:
                ref[o:o + p.numel()] += q2.grad.reshape(-1) / world
    # second step: two of the weights take the direct route the HIP autograd Functions use (the gradient is computed
    # straight into the flat buffer through the parameter's sink and autograd gets None for it); the rest accumulate
    # through autograd hooks.  Same buckets, same result.
    class DirectLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            sink = getattr(w, "_lbmk_grad_sink", None)
            assert sink is not None and sink.active
            sink.write([w], lambda out: out.copy_(dy.t() @ x))
            return dy @ w, None, dy.sum(0)

    red.begin_step()
    h = torch.relu(DirectLinear.apply(x, net[0].weight, net[0].bias))
    h = DirectLinear.apply(h, net[2].weight, net[2].bias)
    net[3](h).pow(2).sum().backward()
    n_direct = len(red.written)
    scale = red.finish()[0]
    err_direct = float((grp.grad * scale - ref).abs().max())
    assert not red.active
    q.put((rank, len(grp.buckets), float((flat_mean - ref).abs().max()), bool(all(p.data_ptr() >= grp.flat.data_ptr() for p in grp.params)),
           n_direct, err_direct))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_world2():
    """GradReducer on CPU tensors over gloo: buckets fire from autograd hooks, unused parameters are still
    reduced, and the averaged flat gradient equals the mean of the per-rank gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbuckets, err, views, n_direct, err_direct in res:
        assert nbuckets >= 3
        assert err < 1e-5
        assert views
        assert n_direct == 2 and err_direct < 1e-5


def _torch_adamw_step(p_flat, g_flat, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, step=1, grad_scale=1.0):
    """CPU stand-in for the fused AdamW kernel (torch.optim.AdamW's update rule on a flat range)"""
    g = g_flat.float() * grad_scale
    p = p_flat.float() * (1 - lr * weight_decay)
    m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    mhat, vhat = m / (1 - betas[0] ** step), v / (1 - betas[1] ** step)
    p_flat.copy_((p - lr * mhat / (vhat.sqrt() + eps)).to(p_flat.dtype))


def _exchange_worker(rank, world, port, q, mode):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys, types
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = types.ModuleType("lbm_amd"); pkg.__path__ = [os.path.join(os.path.dirname(os.path.dirname(__file__)), "lbm_amd")]
    sys.modules.setdefault("lbm_amd", pkg)
    ops = types.ModuleType("lbm_amd.ops")
    ops.adamw_step = _torch_adamw_step
    sys.modules["lbm_amd.ops"] = ops
    sys.modules["lbm_amd"].ops = ops
    from lbm_amd.trainer.trainer import TrainingPipeline
    from lbm_amd.trainer.training_config import TrainingConfig

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(16, 40), torch.nn.ReLU(), torch.nn.Linear(40, 24),
                                            torch.nn.ReLU(), torch.nn.Linear(24, 4))
            self.unused = torch.nn.Parameter(torch.full((11,), 0.5))       # never receives a gradient
            self.register_buffer("stat", torch.randn(3))

        def forward(self, batch):
            return {"loss": self.body(batch["x"]).pow(2).mean()}

    torch.manual_seed(1000 + rank)            # deliberately different initialisation per rank
    net = Net()
    pipe = TrainingPipeline(net, TrainingConfig(learning_rate=1e-2, optimizer_name="AdamW", trainable_params=["body.*", "unused"]),
                            dist=dist, grad_exchange=mode, bucket_bytes=1024)
    pipe.configure_optimizers()
    # after configure_optimizers every replica equals rank 0's initialisation, buffers included
    torch.manual_seed(1000)
    ref = Net()
    same_start = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    opt = torch.optim.AdamW([p for p in ref.parameters()], lr=1e-2)
    losses = []
    for step in range(3):
        xs = [torch.randn(5, 16, generator=torch.Generator().manual_seed(7 * step + r)) for r in range(world)]
        out = pipe.fit_step({"x": xs[rank]})
        losses.append(float(out["loss_mean"]))
        opt.zero_grad(set_to_none=True)
        ref_loss = sum(ref({"x": x})["loss"] for x in xs) / world
        ref_loss.backward()
        opt.step()
        losses[-1] -= float(ref_loss.detach())
    err = max(float((a - b).abs().max()) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    g = pipe.optims[0].groups[0]
    q.put((rank, same_start, err, max(abs(l) for l in losses), float(net.unused.detach().sub(0.5).abs().max()),
           len(g.buckets), g.m.numel(), g.numel))
    dist.destroy_process_group()


def _run_exchange(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_reduce_scatter_all_gather_training_world2():
    """three data-parallel AdamW steps through TrainingPipeline.fit_step on gloo with the reduce-scatter -> sharded
    AdamW -> all-gather exchange: replicas are broadcast from rank 0, end equal to a single-process torch.optim.AdamW
    on the mean gradient, the parameter that never gets a gradient is left alone (no weight decay), the logged loss
    is the mean over ranks, and every rank keeps moments for its half of the buffer only"""
    for rank, same_start, err, loss_err, unused_drift, nbuckets, m_numel, numel in _run_exchange("reduce_scatter"):
        assert same_start
        assert err < 1e-5, err
        assert loss_err < 1e-6
        assert unused_drift == 0.0
        assert nbuckets >= 2 and m_numel * 2 == numel


def test_all_reduce_training_world2():
    """same steps with whole-buffer all-reduce (the switchable fallback)"""
    for rank, same_start, err, loss_err, unused_drift, nbuckets, m_numel, numel in _run_exchange("all_reduce"):
        assert same_start and err < 1e-5 and loss_err < 1e-6 and unused_drift == 0.0
        assert m_numel == numel


def _ragged_worker(rank, world, port, q, tmp):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys, types
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = types.ModuleType("lbm_amd"); pkg.__path__ = [os.path.join(os.path.dirname(os.path.dirname(__file__)), "lbm_amd")]
    sys.modules.setdefault("lbm_amd", pkg)
    ops = types.ModuleType("lbm_amd.ops")
    ops.adamw_step = _torch_adamw_step
    sys.modules["lbm_amd.ops"] = ops
    sys.modules["lbm_amd"].ops = ops
    from lbm_amd.trainer.trainer import TrainingPipeline
    from lbm_amd.trainer.training_config import TrainingConfig

    class Net(torch.nn.Module):
        """a shared trunk and two heads; which head a step uses is decided per call (per rank in this test)"""
        def __init__(self):
            super().__init__()
            self.trunk = torch.nn.Sequential(torch.nn.Linear(16, 48), torch.nn.ReLU())
            self.head_a = torch.nn.Sequential(torch.nn.Linear(48, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
            self.head_b = torch.nn.Sequential(torch.nn.Linear(48, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))

        def forward(self, batch):
            head = self.head_a if batch["head"] == 0 else self.head_b
            return {"loss": head(self.trunk(batch["x"])).pow(2).mean()}

    torch.manual_seed(2000)
    net = Net()
    cfg = TrainingConfig(learning_rate=1e-2, optimizer_name="AdamW", trainable_params=["trunk.*", "head_a.*", "head_b.*"])
    pipe = TrainingPipeline(net, cfg, dist=dist, grad_exchange="reduce_scatter", bucket_bytes=512)
    pipe.configure_optimizers()
    torch.manual_seed(2000)
    ref = Net()
    opt = torch.optim.AdamW(list(ref.parameters()), lr=1e-2)
    order_log = []
    send = pipe.reducer._send
    pipe.reducer._send = lambda g, b: (order_log.append((pipe.reducer._gindex[id(g)], b)), send(g, b))[1]
    for step in range(4):
        xs = [torch.randn(5, 16, generator=torch.Generator().manual_seed(11 * step + r)) for r in range(world)]
        # step 0: both ranks use head_a, so head_b gets no gradient anywhere and joins one step late (its AdamW bias
        # correction must then start at step 1, as torch.optim.AdamW's per-parameter state does); afterwards rank r
        # uses head r: head_a's parameters get no gradient on rank 1, head_b's none on rank 0
        heads = [0 if step == 0 else r for r in range(world)]
        pipe.fit_step({"x": xs[rank], "head": heads[rank]})
        opt.zero_grad(set_to_none=True)
        (sum(ref({"x": x, "head": heads[r]})["loss"] for r, x in enumerate(xs)) / world).backward()
        opt.step()
    err = max(float((a - b).abs().max()) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    nb = [len(g.buckets) for g in pipe.optims[0].groups]
    # a checkpoint written by this 2-rank reduce-scatter run (sharded moments, buckets padded to 16 elements) ...
    path = os.path.join(tmp, "two_ranks.ckpt")
    pipe.save_checkpoint(path, write=rank == 0)
    dist.barrier()
    ck_err, step4_err = -1.0, -1.0
    if rank == 0:
        # ... resumes in ONE process with whole-buffer moments (different padding, different flat length)
        torch.manual_seed(1)
        net1 = Net()
        solo = TrainingPipeline(net1, cfg, dist=None, bucket_bytes=4096)
        solo.load_checkpoint(path)
        assert [g.numel for g in solo.optims[0].groups] != [g.numel for g in pipe.optims[0].groups] or nb != [len(g.buckets) for g in solo.optims[0].groups]
        ck_err = 0.0
        names = {id(p): n for n, p in net1.named_parameters()}
        ref_p = dict(ref.named_parameters())
        for g in solo.optims[0].groups:
            for p, o in zip(g.params, g.offsets):
                st = opt.state[ref_p[names[id(p)]]]
                ck_err = max(ck_err, float((g.m[o:o + p.numel()] - st["exp_avg"].reshape(-1)).abs().max()),
                             float((g.v[o:o + p.numel()] - st["exp_avg_sq"].reshape(-1)).abs().max()))
        x4 = torch.randn(5, 16, generator=torch.Generator().manual_seed(99))
        solo.fit_step({"x": x4, "head": 0})
        # torch.optim.AdamW keeps a step count per parameter: head_b's (untouched now) is left alone, the others advance
        opt.zero_grad(set_to_none=True)
        ref({"x": x4, "head": 0})["loss"].backward()
        opt.step()
        step4_err = max(float((a - b).abs().max()) for a, b in zip(net1.state_dict().values(), ref.state_dict().values()))
        # a pipeline with other trainable groups refuses the file BEFORE touching its weights
        torch.manual_seed(5)
        net2 = Net()
        before = {k: v.clone() for k, v in net2.state_dict().items()}
        other = TrainingPipeline(net2, TrainingConfig(learning_rate=1e-2, optimizer_name="AdamW", trainable_params=["trunk.*"]), dist=None)
        try:
            other.load_checkpoint(path)
            refused = False
        except RuntimeError:
            refused = True
        assert refused and all(torch.equal(before[k], v) for k, v in net2.state_dict().items())
    q.put((rank, err, nb, order_log, ck_err, step4_err))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_touch_different_parameters_and_checkpoint_is_layout_free(tmp_path):
    """Rank 0 trains head_a, rank 1 head_b (each leaves the other head's buckets incomplete during backward): both ranks
    still issue their bucket collectives in the same order, the result equals single-process AdamW on the mean gradient,
    and the checkpoint this 2-rank reduce-scatter run writes resumes in one process with another bucket layout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, err0, nb0, order0, ck_err, step4_err), (_, err1, nb1, order1, _, _) = res
    assert err0 < 1e-5 and err1 < 1e-5, (err0, err1)
    assert sum(nb0) >= 5 and max(nb0) >= 2 and nb0 == nb1
    assert order0 == order1 and len(order0) == 4 * sum(nb0)         # same collectives, same order, on both ranks
    assert 0 <= ck_err < 1e-6, ck_err
    assert 0 <= step4_err < 1e-5, step4_err
