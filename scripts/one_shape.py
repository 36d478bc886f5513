"""Runs one implicit-GEMM shape with one tile a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
tile = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "conv512"
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
if kind == "conv512":
    B, H, cin, cout = 16, 128, 512, 512
elif kind == "conv128":
    B, H, cin, cout = 4, 1024, 128, 128
elif kind == "conv512big":      # the dominant benchmark shape: 1.07 GB in, 1.07 GB out (past the 256 MB cache)
    B, H, cin, cout = 16, 256, 512, 512
x = torch.randn(B, H, H, cin, device=dev, generator=g).to(BF)
w = torch.randn(cout, cin, 3, 3, device=dev, generator=g).to(BF) * (9 * cin) ** -0.5
pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
for _ in range(3):
    y = ops.conv2d(x, pw, tile=tile)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
