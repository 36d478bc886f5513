"""Every distinct implicit-GEMM launch of the headline inference step (1024x1024, batch 16), timed IN PLACE with HIP
events around each launch of a real step: count, total time, TFLOP/s of the executed products.  One JSON line per
shape, heaviest first."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
res, B = 1024, int(os.environ.get("SHAPES_BATCH", "16"))
model = bench.build_model(dev, res)
g = torch.Generator().manual_seed(1)
imgs = (torch.rand(B, 3, res, res, generator=g) * 2 - 1).to(BF).to(dev)
eps = torch.randn(B, 4, res // 8, res // 8, generator=g).to(BF).to(dev)
for _ in range(2):
    bench.one_step(model, imgs, eps, 1)
torch.cuda.synchronize()
recs = []
orig = ops._igemm
def patched(desc, alg_flops=None, executed=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig(desc, alg_flops, executed)
    b.record()
    key = (desc.M, desc.N, desc.K, desc.KH, desc.stride, desc.up2, desc.out_up2 and 1, desc.batch, desc.act, desc.out_f32,
           1 if desc.gn_partial else 0, 1 if desc.residual else 0)
    recs.append((key, a, b, 2.0 * desc.M * desc.N * desc.K * max(desc.batch, 1)))
ops._igemm = patched
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
bench.one_step(model, imgs, eps, 1)
t1.record()
torch.cuda.synchronize()
ops._igemm = orig
agg = collections.OrderedDict()
for key, a, b, fl in recs:
    e = agg.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1; e[1] += a.elapsed_time(b); e[2] += fl
rows = []
for (M, N, K, KH, s, up, oup, z, act, f32, gn, resid), (count, ms, fl) in agg.items():
    rows.append({"M": M, "N": N, "K": K, "KH": KH, "stride": s, "up2": up, "out_up2": oup, "batch": z, "act": act, "f32": f32, "gn": gn,
                 "res": resid, "count": count, "ms_total": round(ms, 3), "tflops": round(fl / ms / 1e9)})
rows.sort(key=lambda r: -r["ms_total"])
print(json.dumps({"step_ms": round(t0.elapsed_time(t1), 1), "igemm_ms": round(sum(r["ms_total"] for r in rows), 1), "launches": len(recs), "shapes": len(rows)}))
for r in rows:
    print(json.dumps(r))
