"""Training pipeline with the interface of the reference's TrainingPipeline
(src/lbm/trainer/trainer.py:15-270) without PyTorch Lightning (not installed here): the same
constructor, ``configure_optimizers`` (regex-selected trainable parameters, 10x learning rate for
``conditioner.*``), ``training_step`` and ``log_samples``; ``fit_step`` is the piece Lightning's loop and
its DDP strategy provide in the reference (zero grads -> training_step -> backward -> gradient
all-reduce -> optimizer step).

MI355X design of the exchange step: every trainable parameter of a group is a view into ONE flat bf16
buffer and its gradient a view into a second one, so the data-parallel all-reduce runs on large
contiguous slices (buckets) of the gradient buffer with no packing copies.  A bucket is reduced (RCCL,
asynchronously) as soon as autograd has produced all of its gradients, i.e. underneath the rest of the
backward pass; buckets are filled in reverse registration order because the last layers' gradients
arrive first.  xGMI is point-to-point, a ring all-reduce is bound by one link (~153 GB/s), so buckets
are large (128 MB default).  The optimizer is one fused AdamW kernel per group over the flat buffers
(f32 moments); the 1/world_size average is folded into it.
"""
import importlib
import logging
import os
import re
import time
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from .training_config import TrainingConfig

BUCKET_BYTES = 128 << 20


def merge_ranges(ranges):
    """sorted, coalesced copy of a list of [start, end) ranges"""
    out = []
    for s, e in sorted(r for r in ranges if r[1] > r[0]):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return [tuple(r) for r in out]


def intersect_ranges(a, b):
    """intersection of two sorted, coalesced range lists"""
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append((s, e))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def _plain(obj):
    """nested dict / list / scalar copy of a hyper-parameter record (anything else by its repr), so that the
    checkpoint stays loadable with ``torch.load(..., weights_only=True)``"""
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if hasattr(obj, "to_dict"):
        return _plain(obj.to_dict())
    if hasattr(obj, "__dict__") and not isinstance(obj, type):
        return _plain({k: v for k, v in vars(obj).items() if not k.startswith("_")})
    return repr(obj)


class FlatGroup:
    """Parameters of one optimizer group re-homed into flat buffers (parameters become views).

    ``shards`` > 1 lays the buckets out for a reduce-scatter / all-gather exchange: every bucket's length is a multiple
    of 8 * shards elements (zero padding at its tail), so rank r owns the r-th equal slice of every bucket, each slice
    starting on a 16-byte boundary."""

    def __init__(self, params: List[nn.Parameter], lr: float, bucket_bytes: int = BUCKET_BYTES, shards: int = 1):
        self.params = list(params)[::-1]          # reverse registration order ~ order gradients become ready
        self.lr = lr
        self.shards = max(1, int(shards))
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), "one device / dtype per group"
        per = max(1, bucket_bytes // torch.empty((), dtype=dt).element_size())
        unit = 8 * self.shards
        # buckets: contiguous [start, end) ranges of the flat buffers, each covering whole parameters
        offs, n = [], 0
        self.buckets, self.param_bucket = [], []
        start, count = 0, 0
        for i, p in enumerate(self.params):
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8          # keep every view 16-byte aligned
            self.param_bucket.append(len(self.buckets))
            count += 1
            if n - start >= per or i + 1 == len(self.params):
                n = (n + unit - 1) // unit * unit
                self.buckets.append((start, n, count))
                start, count = n, 0
        self.numel = n
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.grad = torch.zeros(n, dtype=dt, device=dev)
        for p, o in zip(self.params, offs):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.offsets = offs
        self.m = self.v = None

    def rebind_grads(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.grad[o:o + p.numel()].view_as(p)

    def shard_of(self, bucket: int, rank: int):
        """[start, end) of rank's slice of a bucket (the whole bucket when unsharded)"""
        s, e, _ = self.buckets[bucket]
        step = (e - s) // self.shards
        return s + rank * step, s + (rank + 1) * step

    def moment_offset(self, pos: int, bucket: int, rank: int) -> int:
        """index into this rank's moment buffers of flat position ``pos`` inside its slice of ``bucket``: the slices of
        consecutive buckets are stored back to back, so the buffers hold numel / shards elements"""
        s, _, _ = self.buckets[bucket]
        return s // self.shards + (pos - self.shard_of(bucket, rank)[0])


class GradSink:
    """Direct route for weight gradients: the autograd Functions compute dW straight into the parameter's slot
    of the flat gradient buffer instead of handing autograd a temporary to accumulate.  Active only inside a
    trainer step (between ``begin_step`` and ``finish``); a parameter is expected to be used once per step
    (a second write accumulates, which is only valid while its bucket has not been sent)."""

    def __init__(self, reducer, group):
        self.reducer, self.group = reducer, group
        self.slot = {id(p): (o, i) for i, (p, o) in enumerate(zip(group.params, group.offsets))}

    @property
    def active(self):
        return self.reducer.active

    def stacked(self, params, stack) -> bool:
        """True when ``stack`` (a [rows, K] view over the parameters' memory) is one block of the flat buffer"""
        first = min(params, key=lambda p: p.data_ptr())
        return (all(id(p) in self.slot for p in params) and stack.data_ptr() == first.data_ptr()
                and stack.untyped_storage().data_ptr() == self.group.flat.untyped_storage().data_ptr()
                and stack.numel() == sum(p.numel() for p in params))

    def write(self, params, compute, reads=()):
        """compute(out) fills ``out``: the gradient slot of ``params`` (several = one contiguous block, lowest
        address first).  ``reads``: the tensors ``compute`` consumes (kept alive for the side stream when the
        reducer runs weight gradients on one, see GradReducer.use_side_stream)."""
        first = min(params, key=lambda p: self.slot[id(p)][0])
        o = self.slot[id(first)][0]
        n = sum(p.numel() for p in params)
        rows = sum(p.shape[0] for p in params)
        out = self.group.grad[o:o + n].view(rows, *first.shape[1:])
        fresh = [id(p) not in self.reducer.written for p in params]

        def run():
            if all(fresh):
                compute(out)
            elif not any(fresh):
                tmp = torch.empty_like(out)
                compute(tmp)
                out.add_(tmp)
            else:
                raise RuntimeError("stacked parameters must be written together")
            self._mark(params, fresh)

        self.reducer.on_side_stream(run, reads)

    def slots(self, params):
        """Views of the gradient slots of ``params`` (each shaped like its parameter) for a producer kernel to write its
        result straight into -- bias and norm scale / shift gradients, which autograd would otherwise add to the zeroed
        slot with one small kernel each -- or None when one of them already holds a gradient this step (the caller then
        hands its gradient to autograd, which accumulates).  Call ``commit`` once the producer has been launched."""
        if any(id(p) not in self.slot or id(p) in self.reducer.written for p in params):
            return None
        return [self.group.grad[self.slot[id(p)][0]:self.slot[id(p)][0] + p.numel()].view_as(p) for p in params]

    def commit(self, params):
        self._mark(params, [True] * len(params))

    def _mark(self, params, fresh):
        for p, f in zip(params, fresh):
            if f:
                self.reducer.written.add(id(p))
                self.reducer.param_ready(self.group, self.slot[id(p)][1])
            elif self.reducer.dist is not None:
                raise RuntimeError("a parameter used twice per step cannot take the direct gradient route under data parallelism")


class GradReducer:
    """Bucketed, overlapped data-parallel gradient exchange over the flat gradient buffers.

    ``mode="reduce_scatter"`` (default under data parallelism): a bucket is reduce-scattered as soon as its gradients
    exist, so rank r ends the backward pass holding the summed gradient of ITS slice of every bucket; the optimizer
    then updates only that slice (its moments exist only there) and the updated parameters are all-gathered.  On the
    point-to-point xGMI mesh each phase moves (N-1)/N of the bytes once, where a ring all-reduce moves them twice
    through one link, and the optimizer pass shrinks N-fold.  ``mode="all_reduce"`` keeps whole-buffer replicas.

    The reducer also records which parameters received a gradient in the step (``touched``): like
    ``torch.optim.AdamW``, the optimizer leaves the others alone (no weight decay, no moment decay)."""

    def __init__(self, groups: List[FlatGroup], dist=None, direct_grads: bool = True, mode: str = "reduce_scatter"):
        assert mode in ("reduce_scatter", "all_reduce")
        self.groups, self.dist = groups, dist
        self.mode = mode if dist is not None else "all_reduce"
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        if self.mode == "reduce_scatter":
            assert all(g.shards == self.world for g in groups), "FlatGroup(shards=world_size) is required"
        self.handles = []
        self._pending = {}
        self.active = False
        self.written = set()
        self.touched = [set() for _ in groups]
        # optional weight-gradient side stream (LBM_WGRAD_STREAM=1).  Off by default: measured on the 512x512 batch-8
        # step it is 4-5 % SLOWER (172 vs 164 ms) -- nearly every kernel of the step already fills the chip, so two
        # streams have little to overlap and the per-gradient event traffic costs more than the tails it hides
        self._side = None
        self._main = None
        self.use_side_stream = groups[0].flat.is_cuda and os.environ.get("LBM_WGRAD_STREAM", "0") == "1"
        for gi, g in enumerate(groups):
            sink = GradSink(self, g) if direct_grads else None
            for pi, p in enumerate(g.params):
                if sink is not None:
                    p._lbmk_grad_sink = sink
                p.register_post_accumulate_grad_hook(self._make_hook(g, pi))
        self._gindex = {id(g): gi for gi, g in enumerate(groups)}
        # ONE order in which buckets go on the wire, the same on every rank whatever order the gradients arrive in
        # (a collective pairs up by issue order: two ranks sending different buckets at the same position would hang
        # or, with equal sizes, sum one bucket with another).  Buckets fill from the front of each flat buffer
        # (reverse registration order), so that is the order; a bucket whose turn has not come waits, sent or flushed
        # by finish() behind its predecessors.
        self._order = [(gi, b) for gi, g in enumerate(groups) for b in range(len(g.buckets))]
        self._order_pos = {gb: k for k, gb in enumerate(self._order)}
        self._complete = set()
        self._cursor = 0

    def on_side_stream(self, fn, reads=()):
        """Runs ``fn`` (kernel launches) on the side stream, ordered after everything enqueued so far on the current
        stream; ``reads``: tensors ``fn`` consumes, kept from being recycled until the side stream is done with them."""
        if not self.use_side_stream:
            return fn()
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=main.device)
        self._main = main
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            fn()
        for t in reads:
            if t is not None and t.is_cuda:
                t.record_stream(self._side)

    def _join_streams(self):
        """the current stream waits for the backward pass's main and side streams (gradients are written on both)"""
        if self._side is None:
            return
        cur = torch.cuda.current_stream()
        for st in (self._side, self._main):
            if st is not None and st != cur:
                cur.wait_stream(st)

    def _send(self, group, bucket):
        self._join_streams()
        s, e, _ = group.buckets[bucket]
        if self.mode == "reduce_scatter":
            a, b = group.shard_of(bucket, self.rank)
            self.handles.append(self.dist.reduce_scatter_tensor(group.grad[a:b], group.grad[s:e],
                                                                op=self.dist.ReduceOp.SUM, async_op=True))
        else:
            self.handles.append(self.dist.all_reduce(group.grad[s:e], op=self.dist.ReduceOp.SUM, async_op=True))

    def param_ready(self, group, index):
        """parameter ``index`` of ``group`` holds its final gradient: send its bucket when that is complete"""
        if not self.active:
            return
        self.touched[self._gindex[id(group)]].add(index)
        if self.dist is None:
            return
        bucket = group.param_bucket[index]
        left = self._pending[id(group)]
        left[bucket] -= 1
        if left[bucket] == 0:
            self._complete.add((self._gindex[id(group)], bucket))
            self._send_in_order(flush=False)

    def _send_in_order(self, flush: bool):
        """sends the buckets whose turn has come: the next of the fixed order while it is complete (all of the rest
        when flushing at the end of the step)"""
        while self._cursor < len(self._order):
            gi, b = self._order[self._cursor]
            if not flush and (gi, b) not in self._complete:
                return
            self._send(self.groups[gi], b)
            self._cursor += 1

    def _make_hook(self, group, index):
        def hook(_param):
            self.param_ready(group, index)
        return hook

    def begin_step(self):
        self.handles = []
        self.written = set()
        self.touched = [set() for _ in self.groups]
        self._pending = {id(g): [c for (_, _, c) in g.buckets] for g in self.groups}
        self._complete, self._cursor = set(), 0
        for g in self.groups:
            g.grad.zero_()
            g.rebind_grads()
        self.active = True

    def finish(self, loss: Optional[torch.Tensor] = None):
        """Sends what is left, waits for every bucket, and agrees across ranks on which parameters were touched.
        Returns (factor that turns the summed gradient into the mean, per-group update ranges in flat coordinates
        restricted to this rank's slices, loss averaged over the ranks or None)."""
        self.active = False
        self._join_streams()
        mean_loss = None if loss is None else loss.detach().float()
        if self.dist is not None:
            # parameters that received no gradient this step never fired their hook: their buckets, and every bucket
            # waiting behind them, go out now, in the fixed order
            self._send_in_order(flush=True)
            # one small vector: a flag per parameter (received a gradient on this rank) + the local loss
            # (the reference logs the loss with sync_dist=True, src/lbm/trainer/trainer.py:220-227)
            dev = self.groups[0].flat.device
            flags = torch.zeros(sum(len(g.params) for g in self.groups) + 1, dtype=torch.float32)
            base = 0
            for gi, g in enumerate(self.groups):
                for i in self.touched[gi]:
                    flags[base + i] = 1.0
                base += len(g.params)
            flags = flags.to(dev)
            if mean_loss is not None:
                flags[-1] = mean_loss.reshape(()).to(dev)
            self.dist.all_reduce(flags, op=self.dist.ReduceOp.SUM)
            host = flags.cpu()
            base = 0
            for gi, g in enumerate(self.groups):
                self.touched[gi] = {i for i in range(len(g.params)) if host[base + i] > 0}
                base += len(g.params)
            if mean_loss is not None:
                mean_loss = host[-1] / self.world
            for h in self.handles:
                h.wait()
        ranges = []
        for gi, g in enumerate(self.groups):
            if len(self.touched[gi]) == len(g.params):
                mine = [(0, g.numel)]
            else:
                mine = merge_ranges([(g.offsets[i], g.offsets[i] + (g.params[i].numel() + 7) // 8 * 8)
                                     for i in self.touched[gi]])
            if self.mode == "reduce_scatter":
                mine = intersect_ranges(mine, [g.shard_of(b, self.rank) for b in range(len(g.buckets))])
            ranges.append(mine)
        return 1.0 / self.world, ranges, mean_loss

    def gather_params(self):
        """reduce_scatter mode: every rank updated its slices only; collect the others' (in place)"""
        if self.mode != "reduce_scatter":
            return
        hs = []
        for g in self.groups:
            for b, (s, e, _) in enumerate(g.buckets):
                a, c = g.shard_of(b, self.rank)
                hs.append(self.dist.all_gather_into_tensor(g.flat[s:e], g.flat[a:c], async_op=True))
        for h in hs:
            h.wait()

    def broadcast_params(self):
        """replicas start identical: rank 0's flat parameter buffers win (what DDP does at construction)"""
        if self.dist is None:
            return
        for g in self.groups:
            self.dist.broadcast(g.flat, src=0)


class HipAdamW:
    """torch.optim.AdamW semantics on flat buffers: one fused kernel launch per contiguous range to update (the whole
    buffer when every parameter received a gradient and the buffers are replicated).  With ``shards`` > 1 the moment
    buffers hold only this rank's slice of every bucket (numel / shards elements)."""

    def __init__(self, groups: List[FlatGroup], betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, rank: int = 0,
                 sharded: bool = False, **unused):
        self.groups, self.betas, self.eps, self.weight_decay = groups, tuple(betas), eps, weight_decay
        self.rank, self.sharded = rank, sharded
        self.step_count = 0
        self._small = None          # 1-D parameters and their f32 shadows (built at the first invalidate)
        self.param_groups = [{"params": g.params, "lr": g.lr} for g in groups]
        for g in groups:
            n = g.numel // g.shards if sharded else g.numel
            g.m = torch.zeros(n, dtype=torch.float32, device=g.flat.device)
            g.v = torch.zeros(n, dtype=torch.float32, device=g.flat.device)
            g.steps = [0] * len(g.params)          # updates each parameter has taken (bias correction)

    def _moment_range(self, g, s, e):
        if not self.sharded:
            return s, e
        # a range handed over by GradReducer.finish lies inside one of this rank's bucket slices
        for b in range(len(g.buckets)):
            a, c = g.shard_of(b, self.rank)
            if a <= s and e <= c:
                o = g.moment_offset(s, b, self.rank)
                return o, o + (e - s)
        raise AssertionError("update range outside this rank's slices")

    def step(self, grad_scale: float = 1.0, ranges=None, touched=None):
        """``ranges``: per group, the flat ranges to update (this rank's slices of the touched parameters); ``touched``:
        per group, the indices of the parameters that received a gradient on some rank.  The bias correction uses a
        step count PER PARAMETER, like torch.optim.AdamW's per-parameter state: a parameter that received no gradient
        in a step does not age, so one that joins late starts at step 1.  While every parameter has been touched in
        every step (the usual case) that is one count and one launch per contiguous range."""
        self.step_count += 1
        for gi, (g, pg) in enumerate(zip(self.groups, self.param_groups)):
            todo = [(0, g.numel)] if ranges is None else ranges[gi]
            idx = range(len(g.params)) if touched is None else touched[gi]
            for i in idx:
                g.steps[i] += 1
            counts = {g.steps[i] for i in idx}
            if len(counts) <= 1:
                plan = [(next(iter(counts)) if counts else self.step_count, todo)]
            else:
                plan = [(c, intersect_ranges(merge_ranges([(g.offsets[i], g.offsets[i] + (g.params[i].numel() + 7) // 8 * 8)
                                                           for i in idx if g.steps[i] == c]), todo)) for c in sorted(counts)]
            for count, rs in plan:
                for s, e in rs:
                    ms, me = self._moment_range(g, s, e)
                    ops.adamw_step(g.flat[s:e], g.grad[s:e], g.m[ms:me], g.v[ms:me], lr=pg["lr"], betas=self.betas,
                                   eps=self.eps, weight_decay=self.weight_decay, step=count, grad_scale=grad_scale)

    def invalidate(self):
        """the kernel (and the all-gather) wrote parameters through raw pointers: invalidate cached weight packs, and
        refresh the f32 shadows of the 1-D parameters (biases, norm scales / shifts: ops.f32_of) with one multi-tensor
        copy instead of one conversion kernel per parameter and use"""
        for g in self.groups:
            for p in g.params:
                torch.autograd.graph.increment_version(p)
        if self._small is None:
            self._small = [] if os.environ.get("LBM_NO_F32_SHADOW") else [p for g in self.groups for p in g.params if p.dim() == 1 and p.dtype != torch.float32]
            self._small32 = [torch.empty(p.shape, dtype=torch.float32, device=p.device) for p in self._small]
        if self._small:
            with torch.no_grad():
                torch._foreach_copy_(self._small32, [p.detach() for p in self._small])
            for p, sh in zip(self._small, self._small32):
                p._lbmk_f32 = (sh, p._version)

    def zero_grad(self, set_to_none: bool = False):
        for g in self.groups:
            g.grad.zero_()

    def state_dict(self):
        return {"step": self.step_count, "m": [g.m for g in self.groups], "v": [g.v for g in self.groups],
                "param_steps": [list(g.steps) for g in self.groups]}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        steps = sd.get("param_steps") or [[sd["step"]] * len(g.params) for g in self.groups]
        for g, m, v, st in zip(self.groups, sd["m"], sd["v"], steps):
            g.m.copy_(m)
            g.v.copy_(v)
            g.steps = [int(x) for x in st]


class TrainingPipeline(nn.Module):
    def __init__(self, model, pipeline_config: TrainingConfig, verbose: bool = False, dist=None,
                 grad_exchange: Optional[str] = None, bucket_bytes: int = BUCKET_BYTES, **kwargs):
        """``dist``: ``torch.distributed`` (initialised) for data-parallel training, else None.  ``grad_exchange``:
        "reduce_scatter" (default; environment LBM_GRAD_EXCHANGE overrides) or "all_reduce" — see GradReducer."""
        super().__init__()
        self.grad_exchange = grad_exchange or os.environ.get("LBM_GRAD_EXCHANGE", "reduce_scatter")
        self.bucket_bytes = bucket_bytes
        self.last_loss_mean: Optional[torch.Tensor] = None
        self.model = model
        self.pipeline_config = pipeline_config
        self.log_samples_model_kwargs = pipeline_config.log_samples_model_kwargs
        self.verbose = verbose
        log_keys = pipeline_config.log_keys
        if isinstance(log_keys, str):
            log_keys = [log_keys]
        self.log_keys = log_keys or []
        self.dist = dist
        self.optims: List[Any] = []
        self.reducer: Optional[GradReducer] = None
        self.global_step = 0
        self.epoch = 0               # epoch the next step belongs to (fit() keeps it; saved in checkpoints)
        self.timer = time.perf_counter()

    # ---- same selection rules as the reference (trainer.py:86-124) --------------------------------
    def configure_optimizers(self):
        lr = self.pipeline_config.learning_rate
        lr_mult = {r"conditioner.*": 10.0}
        patterns = [(rx, re.compile(rx)) for rx in self.pipeline_config.trainable_params]
        groups = {rx: [] for rx, _ in patterns}
        n_params, seen = 0, set()
        for name, p in self.model.named_parameters():
            rx_hit = None
            for rx, pat in patterns:
                if pat.match(name):
                    rx_hit = rx
                    break
            if rx_hit is None:
                p.requires_grad = False
                continue
            if p.requires_grad and id(p) not in seen:
                groups[rx_hit].append(p)
                seen.add(id(p))
                n_params += p.numel()
        logging.info(f"trainable params: {n_params}")
        kwargs = dict(self.pipeline_config.optimizer_kwargs or {})
        if self.pipeline_config.optimizer_name == "AdamW":
            sharded = self.dist is not None and self.grad_exchange == "reduce_scatter"
            world = self.dist.get_world_size() if self.dist is not None else 1
            flat = [FlatGroup(ps, lr * lr_mult.get(rx, 1.0), self.bucket_bytes, shards=world if sharded else 1)
                    for rx, ps in groups.items() if ps]
            self.reducer = GradReducer(flat, self.dist, mode=self.grad_exchange)
            optimizer = HipAdamW(flat, rank=self.reducer.rank, sharded=sharded, **kwargs)
            # replicas must start identical (Lightning's DDP strategy broadcasts rank 0's module state): flat buffers
            # of the trainable groups, then every other parameter and buffer (frozen VAE, num_iterations, Fourier B)
            self.reducer.broadcast_params()
            self._broadcast_rest({id(p) for g in flat for p in g.params})
            optimizer.invalidate()
        else:
            # other torch.optim classes keep torch's own update rule; gradients are still reduced here
            param_groups = [{"params": ps, "lr": lr * lr_mult.get(rx, 1.0)} for rx, ps in groups.items() if ps]
            optimizer = getattr(importlib.import_module("torch.optim"), self.pipeline_config.optimizer_name)(param_groups, **kwargs)
            self.reducer = None
        self.optims = [optimizer]
        return [optimizer]

    def _broadcast_rest(self, skip_ids):
        if self.dist is None:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if id(t) not in skip_ids and t.numel() > 0 and t.device.type != "meta":
                self.dist.broadcast(t.data, src=0)

    def training_step(self, train_batch: Dict[str, Any], batch_idx: int) -> dict:
        model_output = self.model(train_batch)
        return {"loss": model_output["loss"], "batch_idx": batch_idx}

    def fit_step(self, train_batch: Dict[str, Any]) -> dict:
        """One optimisation step: what Lightning's loop + DDP do around ``training_step`` in the reference."""
        if not self.optims:
            self.configure_optimizers()
        opt = self.optims[0]
        if self.reducer is not None:
            self.reducer.begin_step()
        else:
            opt.zero_grad(set_to_none=True)
        try:
            out = self.training_step(train_batch, self.global_step)
            out["loss"].backward()
        except BaseException:
            if self.reducer is not None:
                self.reducer.active = False      # no direct gradient writes outside a step
            raise
        if self.reducer is not None:
            scale, ranges, loss_mean = self.reducer.finish(out["loss"])
            opt.step(grad_scale=scale, ranges=ranges, touched=self.reducer.touched)
            self.reducer.gather_params()
            opt.invalidate()
            self.last_loss_mean = loss_mean          # 0-dim tensor (on the host under data parallelism): no extra sync
        else:
            if self.dist is not None:
                world = self.dist.get_world_size()
                for pg in opt.param_groups:
                    for p in pg["params"]:
                        if p.grad is not None:
                            self.dist.all_reduce(p.grad)
                            p.grad.div_(world)
            opt.step()
        self.global_step += 1
        if hasattr(self.model, "on_train_batch_end"):
            self.model.on_train_batch_end(train_batch)
        # "loss_mean": the step's loss averaged over the data-parallel ranks (the reference logs train/loss with
        # sync_dist=True); equals the local loss in a single process
        return {"loss": out["loss"].detach(), "batch_idx": out["batch_idx"], "loss_mean": self.last_loss_mean}

    # ---- checkpoints (what Lightning's ModelCheckpoint / resume do in examples/training/train_base.py:27-53) ----
    def save_checkpoint(self, path: str, write: bool = True):
        """Lightning-layout ``.ckpt``: ``state_dict`` keys carry the ``model.`` prefix (so ``get_model`` loads the
        file), plus the optimizer moments and the step counter.  Tensors only: loadable with weights_only=True.
        Under data parallelism with sharded moments every rank must call this (the moments are gathered);
        ``write=False`` on the ranks that should not write the file."""
        ckpt = {"state_dict": {"model." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "global_step": self.global_step, "epoch": self.epoch}
        if getattr(self, "hparams", None):
            ckpt["hyper_parameters"] = _plain(self.hparams)     # what Lightning's save_hyperparameters keeps in the file
        if self.optims:
            opt = self.optims[0]
            sd = opt.state_dict()
            if isinstance(opt, HipAdamW):
                # moments as the unpadded concatenation of the group's parameters (in the group's order): no trace of
                # the bucket padding, which depends on the rank count and the exchange mode
                ckpt["optimizer_states"] = [{"step": sd["step"],
                                             "m": [self._param_coords(g, self._full_moments(g, g.m)).cpu() for g in opt.groups],
                                             "v": [self._param_coords(g, self._full_moments(g, g.v)).cpu() for g in opt.groups],
                                             "param_numel": [[p.numel() for p in g.params] for g in opt.groups],
                                             "param_steps": sd["param_steps"]}]
            else:
                ckpt["optimizer_states"] = [sd]
        if write:
            torch.save(ckpt, path)

    def _sharded(self) -> bool:
        return self.reducer is not None and self.reducer.mode == "reduce_scatter"

    def _full_moments(self, g: "FlatGroup", mine: torch.Tensor) -> torch.Tensor:
        """moment buffer in flat-parameter coordinates (checkpoints are independent of the exchange mode and of the
        number of ranks that wrote them); every rank must call this when the moments are sharded"""
        if not self._sharded():
            return mine
        full = torch.zeros(g.numel, dtype=mine.dtype, device=mine.device)
        rank = self.reducer.rank
        for b, (s, e, _) in enumerate(g.buckets):
            a, c = g.shard_of(b, rank)
            o = g.moment_offset(a, b, rank)
            full[a:c] = mine[o:o + (c - a)]
            self.dist.all_gather_into_tensor(full[s:e], full[a:c].clone())
        return full

    @staticmethod
    def _param_coords(g: "FlatGroup", full: torch.Tensor) -> torch.Tensor:
        """flat-buffer coordinates (8-element parameter alignment, bucket tail padding) -> the parameters back to back"""
        return torch.cat([full[o:o + p.numel()] for p, o in zip(g.params, g.offsets)])

    @staticmethod
    def _flat_coords(g: "FlatGroup", packed: torch.Tensor) -> torch.Tensor:
        """inverse of _param_coords for THIS run's layout (padding positions hold zeros, as after construction)"""
        full = torch.zeros(g.numel, dtype=packed.dtype, device=packed.device)
        at = 0
        for p, o in zip(g.params, g.offsets):
            full[o:o + p.numel()] = packed[at:at + p.numel()]
            at += p.numel()
        return full

    def _my_moments(self, g: "FlatGroup", full: torch.Tensor) -> torch.Tensor:
        if not self._sharded():
            return full
        rank = self.reducer.rank
        return torch.cat([full[slice(*g.shard_of(b, rank))] for b in range(len(g.buckets))])

    def load_checkpoint(self, path: str, strict: bool = True):
        """Restores parameters in place (they stay views of the flat buffers), optimizer state and the step."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in ckpt["state_dict"].items()}
        own = self.model.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"checkpoint mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
        states = ckpt.get("optimizer_states")
        if states:
            # checked BEFORE any weight is overwritten: a refused checkpoint leaves the pipeline as it was
            if not self.optims:
                self.configure_optimizers()
            if isinstance(self.optims[0], HipAdamW):
                want = [[p.numel() for p in g.params] for g in self.optims[0].groups]
                if states[0].get("param_numel") != want:
                    raise RuntimeError("optimizer state of the checkpoint does not match this pipeline's trainable "
                                       "parameter groups (trainable_params / model differ from the run that wrote it)")
            own = self.model.state_dict()        # configure_optimizers re-homes parameters into the flat buffers
        with torch.no_grad():
            for k, t in own.items():
                if k in sd:
                    t.copy_(sd[k].to(t.dtype))
        for p in self.model.parameters():
            torch.autograd.graph.increment_version(p)
        self.global_step = int(ckpt.get("global_step", 0))
        self.epoch = int(ckpt.get("epoch", 0))
        if states:
            opt = self.optims[0]
            if isinstance(opt, HipAdamW):
                st = states[0]
                opt.load_state_dict({"step": st["step"], "param_steps": st.get("param_steps"),
                                     "m": [self._my_moments(g, self._flat_coords(g, t)) for g, t in zip(opt.groups, st["m"])],
                                     "v": [self._my_moments(g, self._flat_coords(g, t)) for g, t in zip(opt.groups, st["v"])]})
            else:
                opt.load_state_dict(states[0])

    def validation_step(self, val_batch: Dict[str, Any], val_idx: int) -> dict:
        with torch.no_grad():
            loss = self.model(val_batch)["loss"]
        return {"loss": loss, "metrics": {}}

    def log_samples(self, batch: Dict[str, Any]):
        logs = self.model.log_samples(batch, **self.log_samples_model_kwargs)
        N = min([logs[k].shape[0] for k in logs]) if logs else 0
        for key in self.log_keys:
            if key in batch:
                logs[key] = batch[key][:N] if N > 0 else batch[key]
        return logs
