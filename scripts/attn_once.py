"""A few launches of the d=64 flash kernel on the benchmark's 64x64-latent shape (for counter runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
B, H, N, D = 16, 10, 4096, 64
q = torch.randn(B, N, H * D, device=dev).bfloat16()
k = torch.randn(B, N, H * D, device=dev).bfloat16()
vt = torch.randn(B, H * D, N, device=dev).bfloat16()
for _ in range(3):
    ops.flash_attn(q, k, vt, H, D)
torch.cuda.synchronize()
