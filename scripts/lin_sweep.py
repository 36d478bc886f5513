"""Sweep of small linear-layer GEMMs over the tile table; one JSON line per shape (times in us)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
tiles = [-1, 0, 1, 3, 7, 10, 11, 12, 13, 14]
shapes = [(M, N, K) for M in (1024, 2048, 4096, 8192, 16384) for N in (320, 640, 1280, 2560, 5120, 10240) for K in (640, 1280, 5120)
          if M * N <= 2048 * 10240 * 2 and not (N == 320 and K == 5120)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16() * K ** -0.5
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev).bfloat16())
    res = {}
    for t in tiles:
        for _ in range(5):
            ops.linear(x, pw, tile=t)
    for rep in range(2):
        for t in tiles:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                ops.linear(x, pw, tile=t)
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 30 * 1e3
            res[t] = min(res.get(t, 1e9), us)
    print(json.dumps({"M": M, "N": N, "K": K, "us": {str(t): round(v, 1) for t, v in res.items()}}), flush=True)
