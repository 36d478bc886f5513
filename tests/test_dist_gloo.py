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
