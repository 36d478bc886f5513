"""Fused attention backward timing: (B, N, H, D) cases of both UNet configurations at 512^2 / 1024^2."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops, _C
dev = torch.device("cuda:0")
cases = [(8, 1024, 10, 64), (8, 256, 20, 64), (8, 4096, 8, 40), (8, 1024, 8, 80), (8, 256, 8, 160), (4, 4096, 10, 64), (4, 4096, 8, 80), (4, 1024, 8, 160)]
for split in (0,):
    for B, N, H, D in cases:
        C = H * D
        q, k, v, do = (torch.randn(B, N, C, device=dev).bfloat16() for _ in range(4))
        o, lse = ops.flash_attn(q, k, ops.transpose(v), H, D, D ** -0.5, return_lse=True)
        for _ in range(3):
            ops.flash_attn_bwd(q, k, v, o, do, lse, H, D, D ** -0.5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.flash_attn_bwd(q, k, v, o, do, lse, H, D, D ** -0.5)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        fl = 2.0 * B * H * N * N * D * 5      # S, dP, dV, dQ, dK (the second S recompute not counted)
        a.record()
        for _ in range(10):
            ops.flash_attn(q, k, ops.transpose(v), H, D, D ** -0.5)
        b.record(); torch.cuda.synchronize()
        fms = a.elapsed_time(b) / 10
        print(f"B{B} N{N} H{H} D{D}: bwd {ms*1e3:8.1f} us ({fl/ms/1e9:5.0f} TF)   fwd+transpose {fms*1e3:8.1f} us", flush=True)
