"""Replay every distinct implicit-GEMM launch of one benchmark step with each tile shape; one JSON line per shape.
Descriptors are replayed as recorded (their buffers stay mapped in torch's caching allocator; outputs are scratch)."""
import os, sys, json, copy, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd import ops, _C

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tiles = [int(t) for t in sys.argv[3].split(',')] if len(sys.argv) > 3 else [-1, 0, 1, 3, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
dev = torch.device("cuda:0")
model = bench.build_model(dev, res)
images = (torch.rand(B, 3, res, res) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, res // 8, res // 8).to(torch.bfloat16).to(dev)
bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()

seen = collections.OrderedDict()
orig = ops._igemm
def patched(desc, alg_flops=None, executed=None):
    orig(desc, alg_flops, executed)
    key = (desc.M, desc.N, desc.K, desc.KH, desc.stride, desc.up2, desc.batch, desc.act, desc.out_f32, 1 if desc.gn_partial else 0,
           desc.Cin, 1 if desc.residual else 0, desc.out_up2)
    if key not in seen:
        d = _C.IgemmDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(desc), ctypes.sizeof(d))
        seen[key] = [d, 0]
    seen[key][1] += 1
ops._igemm = patched
with torch.no_grad():
    keep = bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()
ops._igemm = orig
stream = ops._stream()
for key, (d, count) in seen.items():
    times = {}
    for rep in range(3):
        for t in tiles:
            d.tile = t
            rc = _C.lib.lbmk_igemm(ctypes.byref(d), stream)
            if rc != 0:
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                _C.lib.lbmk_igemm(ctypes.byref(d), stream)
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 3 * 1e3
            times[t] = min(times.get(t, 1e12), us)
    M, N, K, KH, s, up, z, act, f32, gn, cin, resid, oup = key
    print(json.dumps({"M": M, "N": N, "K": K, "KH": KH, "stride": s, "up2": up, "batch": z, "act": act, "f32": f32, "gn": gn, "Cin": cin,
                      "res": resid, "out_up2": oup,
                      "count": count, "us": {str(t): round(v, 1) for t, v in times.items()}}), flush=True)
