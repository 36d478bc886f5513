"""What PyTorch Lightning's ``Trainer.fit`` + ``ModelCheckpoint`` do around ``TrainingPipeline`` in the reference's
training scripts (examples/training/train_base.py:21-53, 103-172), without Lightning: task directory, choice of the
checkpoint to resume from, periodic ``epoch=E-step=S.ckpt`` files, and the epoch / step loop itself.

One process per GPU; under data parallelism every rank runs ``fit`` on its own shard of the data (the loader's
sampler decides which), rank 0 writes the checkpoints.
"""
import logging
from pathlib import Path
from typing import Callable, Iterable, Optional

import torch


def resolve_task_dir(save_ckpt_path: str, task_name: str) -> Path:
    task_dir = Path(save_ckpt_path) / task_name
    task_dir.mkdir(parents=True, exist_ok=True)
    return task_dir


def resolve_resume_checkpoint(task_dir, resume_from_checkpoint: bool, resume_ckpt_path: Optional[str] = None) -> Optional[str]:
    """Resolution order of train_base.py:27-53: nothing unless resuming was asked for; the explicit path when it exists
    (a missing one only warns); ``<task_dir>/last.ckpt``; the most recently modified ``*.ckpt`` of the task directory."""
    if not resume_from_checkpoint:
        return None
    task_dir = Path(task_dir)
    candidates = []
    if resume_ckpt_path:
        if Path(resume_ckpt_path).exists():
            return str(Path(resume_ckpt_path))
        logging.warning("Resume checkpoint not found: %s", resume_ckpt_path)
    candidates.append(task_dir / "last.ckpt")
    candidates += sorted(task_dir.glob("*.ckpt"), key=lambda f: f.stat().st_mtime, reverse=True)
    for c in candidates:
        if c.exists():
            return str(c)
    return None


def checkpoint_name(epoch: int, step: int) -> str:
    """Lightning's default ModelCheckpoint file name"""
    return f"epoch={epoch}-step={step}.ckpt"


def fit(pipeline, train_loader: Iterable, max_epochs: int, save_dir=None, save_interval: Optional[int] = None,
        ckpt_path: Optional[str] = None, to_device: Optional[Callable] = None, on_step: Optional[Callable] = None,
        max_steps: Optional[int] = None) -> int:
    """Runs ``pipeline.fit_step`` over ``train_loader`` for ``max_epochs`` epochs; returns the global step reached.
    ``ckpt_path`` restores parameters, optimizer moments and the step counter first (training continues bit-identically
    from a checkpoint written by ``save_checkpoint``).  Every ``save_interval`` optimisation steps rank 0 writes
    ``save_dir/epoch=E-step=S.ckpt`` (all ranks take part in gathering sharded optimizer state).
    ``on_step(step, out)`` is called after every step (logging hooks)."""
    if ckpt_path:
        logging.info("Resuming training from checkpoint: %s", ckpt_path)
        pipeline.load_checkpoint(ckpt_path)
    rank = pipeline.dist.get_rank() if pipeline.dist is not None else 0
    device = getattr(pipeline.model, "device", None)
    save_dir = None if save_dir is None else Path(save_dir)
    # a resumed run continues in the epoch the checkpoint was written in (the epoch counter only names checkpoints
    # and bounds the loop; like Lightning's mid-epoch resume without a stateful loader, the loader starts over)
    for epoch in range(getattr(pipeline, "epoch", 0), max_epochs):
        pipeline.epoch = epoch
        for batch in train_loader:
            if to_device is not None:
                batch = to_device(batch)
            elif device is not None:
                batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            out = pipeline.fit_step(batch)
            step = pipeline.global_step
            if on_step is not None:
                on_step(step, out)
            if save_dir is not None and save_interval and step % save_interval == 0:
                pipeline.save_checkpoint(str(save_dir / checkpoint_name(epoch, step)), write=rank == 0)
            if max_steps is not None and step >= max_steps:
                return step
    pipeline.epoch = max_epochs
    return pipeline.global_step
