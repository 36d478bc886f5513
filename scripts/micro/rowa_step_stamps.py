"""Shader-clock stamps inside one steady-state K-step of the shared-row tile (diagnostic build, tile id 118):
wave 0 of every workgroup: step start, requests issued, MFMAs issued, own requests landed, barrier passed."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from lbm_amd import ops, _C
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
fn = _C.lib.lbmk_debug_igemm_times
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
for (B, H, cin, cout) in [(4, 1024, 128, 128), (4, 1024, 256, 128), (16, 512, 256, 256)]:
    x = torch.randn(B, H, H, cin, device=dev, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g).bfloat16() * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    for _ in range(4):
        ops.conv2d(x, pw, tile=118)
    torch.cuda.synchronize()
    nb = 256
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, nb) == 0
    t = buf.astype(np.int64)[:, :5]
    seg = np.diff(t, axis=1)
    print(f"conv {cin}->{cout} {H}^2 B{B}: shader cycles, median over {nb} workgroups (p10 / p90)")
    for name, col in zip(("issue requests", "issue MFMAs (+ fragment reads)", "wait for own requests", "barrier"), range(4)):
        v = seg[:, col]
        print(f"  {name:32s} {np.median(v):7.0f}  ({np.percentile(v, 10):.0f} / {np.percentile(v, 90):.0f})")
    tot = t[:, 4] - t[:, 0]
    print(f"  {'whole step':32s} {np.median(tot):7.0f}  ({np.percentile(tot, 10):.0f} / {np.percentile(tot, 90):.0f})   MFMA pipe needs 1536")
