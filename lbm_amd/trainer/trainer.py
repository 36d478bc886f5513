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
import re
import time
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from .training_config import TrainingConfig

BUCKET_BYTES = 128 << 20


class FlatGroup:
    """Parameters of one optimizer group re-homed into flat buffers (parameters become views)."""

    def __init__(self, params: List[nn.Parameter], lr: float, bucket_bytes: int = BUCKET_BYTES):
        self.params = list(params)[::-1]          # reverse registration order ~ order gradients become ready
        self.lr = lr
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), "one device / dtype per group"
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8          # keep every view 16-byte aligned
        self.numel = n
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.grad = torch.zeros(n, dtype=dt, device=dev)
        for p, o in zip(self.params, offs):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.offsets = offs
        # buckets: contiguous [start, end) ranges of the flat gradient, each covering whole parameters
        per = max(1, bucket_bytes // self.flat.element_size())
        self.buckets, self.param_bucket = [], []
        start, count = 0, 0
        for i, (p, o) in enumerate(zip(self.params, offs)):
            self.param_bucket.append(len(self.buckets))
            count += 1
            end = offs[i + 1] if i + 1 < len(offs) else n
            if end - start >= per or i + 1 == len(offs):
                self.buckets.append((start, end, count))
                start, count = end, 0
        self.m = self.v = None

    def rebind_grads(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.grad[o:o + p.numel()].view_as(p)


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

    def write(self, params, compute):
        """compute(out) fills ``out``: the gradient slot of ``params`` (several = one contiguous block, lowest
        address first)."""
        first = min(params, key=lambda p: self.slot[id(p)][0])
        o = self.slot[id(first)][0]
        n = sum(p.numel() for p in params)
        rows = sum(p.shape[0] for p in params)
        out = self.group.grad[o:o + n].view(rows, *first.shape[1:])
        fresh = [id(p) not in self.reducer.written for p in params]
        if all(fresh):
            compute(out)
        elif not any(fresh):
            tmp = torch.empty_like(out)
            compute(tmp)
            out.add_(tmp)
        else:
            raise RuntimeError("stacked parameters must be written together")
        for p, f in zip(params, fresh):
            if f:
                self.reducer.written.add(id(p))
                self.reducer.param_ready(self.group, self.group.param_bucket[self.slot[id(p)][1]])
            elif self.reducer.dist is not None:
                raise RuntimeError("a parameter used twice per step cannot take the direct gradient route under data parallelism")


class GradReducer:
    """Bucketed, overlapped data-parallel gradient all-reduce over the flat gradient buffers."""

    def __init__(self, groups: List[FlatGroup], dist=None, direct_grads: bool = True):
        self.groups, self.dist = groups, dist
        self.handles = []
        self._pending = {}
        self.active = False
        self.written = set()
        for g in groups:
            sink = GradSink(self, g) if direct_grads else None
            for p, b in zip(g.params, g.param_bucket):
                if sink is not None:
                    p._lbmk_grad_sink = sink
                if dist is not None:
                    p.register_post_accumulate_grad_hook(self._make_hook(g, b))

    def param_ready(self, group, bucket):
        """one more parameter of ``bucket`` holds its final gradient: send the bucket when it is complete"""
        if self.dist is None:
            return
        left = self._pending[id(group)]
        left[bucket] -= 1
        if left[bucket] == 0:
            s, e, _ = group.buckets[bucket]
            self.handles.append(self.dist.all_reduce(group.grad[s:e], op=self.dist.ReduceOp.SUM, async_op=True))

    def _make_hook(self, group, bucket):
        def hook(_param):
            self.param_ready(group, bucket)
        return hook

    def begin_step(self):
        self.handles = []
        self.written = set()
        self._pending = {id(g): [c for (_, _, c) in g.buckets] for g in self.groups}
        for g in self.groups:
            g.grad.zero_()
            g.rebind_grads()
        self.active = True

    def finish(self) -> float:
        """Waits for every bucket; returns the factor that turns the summed gradient into the mean."""
        self.active = False
        if self.dist is None:
            return 1.0
        # parameters that received no gradient this step never fired their hook: reduce what is left
        for g in self.groups:
            left = self._pending[id(g)]
            for b, c in enumerate(left):
                if c > 0:
                    s, e, _ = g.buckets[b]
                    self.handles.append(self.dist.all_reduce(g.grad[s:e], op=self.dist.ReduceOp.SUM, async_op=True))
                    left[b] = 0
        for h in self.handles:
            h.wait()
        return 1.0 / self.dist.get_world_size()


class HipAdamW:
    """torch.optim.AdamW semantics on flat buffers, one fused kernel launch per group."""

    def __init__(self, groups: List[FlatGroup], betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **unused):
        self.groups, self.betas, self.eps, self.weight_decay = groups, tuple(betas), eps, weight_decay
        self.step_count = 0
        self.param_groups = [{"params": g.params, "lr": g.lr} for g in groups]
        for g in groups:
            g.m = torch.zeros(g.numel, dtype=torch.float32, device=g.flat.device)
            g.v = torch.zeros(g.numel, dtype=torch.float32, device=g.flat.device)

    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        for g, pg in zip(self.groups, self.param_groups):
            ops.adamw_step(g.flat, g.grad, g.m, g.v, lr=pg["lr"], betas=self.betas, eps=self.eps,
                           weight_decay=self.weight_decay, step=self.step_count, grad_scale=grad_scale)
            for p in g.params:                      # the kernel wrote through raw pointers: invalidate cached packs
                torch.autograd.graph.increment_version(p)

    def zero_grad(self, set_to_none: bool = False):
        for g in self.groups:
            g.grad.zero_()

    def state_dict(self):
        return {"step": self.step_count, "m": [g.m for g in self.groups], "v": [g.v for g in self.groups]}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        for g, m, v in zip(self.groups, sd["m"], sd["v"]):
            g.m.copy_(m)
            g.v.copy_(v)


class TrainingPipeline(nn.Module):
    def __init__(self, model, pipeline_config: TrainingConfig, verbose: bool = False, dist=None, **kwargs):
        super().__init__()
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
            flat = [FlatGroup(ps, lr * lr_mult.get(rx, 1.0)) for rx, ps in groups.items() if ps]
            optimizer = HipAdamW(flat, **kwargs)
            self.reducer = GradReducer(flat, self.dist)
        else:
            # other torch.optim classes keep torch's own update rule; gradients are still reduced here
            param_groups = [{"params": ps, "lr": lr * lr_mult.get(rx, 1.0)} for rx, ps in groups.items() if ps]
            optimizer = getattr(importlib.import_module("torch.optim"), self.pipeline_config.optimizer_name)(param_groups, **kwargs)
            self.reducer = None
        self.optims = [optimizer]
        return [optimizer]

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
            opt.step(grad_scale=self.reducer.finish())
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
        return {"loss": out["loss"].detach(), "batch_idx": out["batch_idx"]}

    # ---- checkpoints (what Lightning's ModelCheckpoint / resume do in examples/training/train_base.py:27-53) ----
    def save_checkpoint(self, path: str):
        """Lightning-layout ``.ckpt``: ``state_dict`` keys carry the ``model.`` prefix (so ``get_model`` loads the
        file), plus the optimizer moments and the step counter.  Tensors only: loadable with weights_only=True."""
        ckpt = {"state_dict": {"model." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "global_step": self.global_step}
        if self.optims:
            opt = self.optims[0]
            sd = opt.state_dict()
            if isinstance(opt, HipAdamW):
                ckpt["optimizer_states"] = [{"step": sd["step"], "m": [t.cpu() for t in sd["m"]], "v": [t.cpu() for t in sd["v"]],
                                             "numel": [g.numel for g in opt.groups]}]
            else:
                ckpt["optimizer_states"] = [sd]
        torch.save(ckpt, path)

    def load_checkpoint(self, path: str, strict: bool = True):
        """Restores parameters in place (they stay views of the flat buffers), optimizer state and the step."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in ckpt["state_dict"].items()}
        own = self.model.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"checkpoint mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
        with torch.no_grad():
            for k, t in own.items():
                if k in sd:
                    t.copy_(sd[k].to(t.dtype))
        for p in self.model.parameters():
            torch.autograd.graph.increment_version(p)
        self.global_step = int(ckpt.get("global_step", 0))
        states = ckpt.get("optimizer_states")
        if states:
            if not self.optims:
                self.configure_optimizers()
            opt = self.optims[0]
            if isinstance(opt, HipAdamW):
                st = states[0]
                assert st["numel"] == [g.numel for g in opt.groups], "optimizer groups differ from the checkpoint's"
                opt.load_state_dict({"step": st["step"], "m": st["m"], "v": st["v"]})
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
