"""One-process-per-GPU helpers.  The inference path shards by image batch: every rank owns a
contiguous slice of the images and never exchanges data with the others; the only
collectives are the timing barrier and the max-over-ranks of the elapsed time.
Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" for CPU tests."""
import os
from typing import Optional, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: Optional[torch.device] = None):
    """Initialises the default process group when WORLD_SIZE > 1; returns torch.distributed or None."""
    rank, _, world = env_world()
    if world <= 1 and os.environ.get("LBM_FORCE_DIST") != "1":     # LBM_FORCE_DIST=1: a one-rank group, to exercise the collectives
        return None
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of ``total`` items owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist, device=None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
