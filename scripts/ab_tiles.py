"""A/B of tile choices on one shape inside one process (same device, sustained clocks): interleaved repeats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
B, H, cin, cout = [int(v) for v in sys.argv[1].split(",")]
tiles = [int(t) for t in sys.argv[2].split(",")]
x = torch.randn(B, H, H, cin, device=dev).bfloat16()
w = torch.randn(cout, cin, 3, 3, device=dev).bfloat16() * (9 * cin) ** -0.5
pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
fl = 2.0 * B * H * H * cout * 9 * cin
for _ in range(30):
    ops.conv2d(x, pw)
res = {t: [] for t in tiles}
for rep in range(4):
    for t in tiles:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv2d(x, pw, tile=t)
        b.record(); torch.cuda.synchronize()
        res[t].append(fl / (a.elapsed_time(b) / 10) / 1e9)
for t in tiles:
    print(f"conv {cin}->{cout} {H}^2 B{B} tile {t:3d}: " + " ".join(f"{v:6.0f}" for v in res[t]))
