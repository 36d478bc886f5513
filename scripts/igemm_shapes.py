"""Per-shape timing of the implicit-GEMM launches of one benchmark step (HIP events per launch)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd import ops

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
model = bench.build_model(dev, res)
images = (torch.rand(B, 3, res, res) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, res // 8, res // 8).to(torch.bfloat16).to(dev)
bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()

records = []
orig = ops._igemm
def patched(desc, alg_flops=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); orig(desc, alg_flops); b.record()
    key = (desc.M, desc.N, desc.K, desc.KH, desc.stride, desc.up2, desc.batch, desc.act, desc.out_f32)
    fl = alg_flops if alg_flops is not None else 2.0 * desc.M * desc.N * desc.K * desc.batch
    records.append((key, fl, a, b))
ops._igemm = patched
bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, a, b in records:
    e = agg.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1; e[1] += a.elapsed_time(b); e[2] += fl
tot = sum(e[1] for e in agg.values())
print(f"total igemm ms {tot:.1f}  launches {len(records)}")
print("     M      N      K kh s up  z act f32 | calls      ms   %tot   TFLOP/s")
for key, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, KH, s, up, z, act, f32 = key
    print(f"{M:9d} {N:6d} {K:6d} {KH:2d} {s} {up} {z:3d} {act} {f32} | {e[0]:5d} {e[1]:8.2f} {100*e[1]/tot:5.1f}  {e[2]/e[1]/1e9:8.1f}")
