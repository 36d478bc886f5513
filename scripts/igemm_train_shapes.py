"""Every distinct implicit-GEMM / gemm_tn launch of one training step (512x512, batch 8, SDXL-style UNet), replayed
in isolation: count, time with the automatic tile, TFLOP/s.  One JSON line per shape, heaviest first."""
import os, sys, json, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd import ops, _C

dev = torch.device("cuda:0")
BF = torch.bfloat16
from lbm_amd.trainer import TrainingConfig, TrainingPipeline
model = bench.build_model(dev, 512)
model.train(); model.vae.freeze()
pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
pipe.configure_optimizers()
g = torch.Generator().manual_seed(1)
batch = {"source_image": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(dev),
         "source_image_paste": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(dev)}
pipe.fit_step(batch); pipe.fit_step(batch)
torch.cuda.synchronize()

seen = collections.OrderedDict()
orig = ops._igemm
def patched(desc, alg_flops=None, executed=None):
    orig(desc, alg_flops, executed)
    key = (desc.M, desc.N, desc.K, desc.KH, desc.stride, desc.up2, desc.batch, desc.act, desc.out_f32, 1 if desc.gn_partial else 0, desc.Cin)
    if key not in seen:
        d = _C.IgemmDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(desc), ctypes.sizeof(d))
        seen[key] = [d, 0]
    seen[key][1] += 1
ops._igemm = patched
tn = collections.OrderedDict()
orig_tn = ops.gemm_tn
def patched_tn(a, b, out_dtype=torch.float32, out=None, splits=None):
    key = (a.shape[0], a.shape[1], b.shape[1])
    tn.setdefault(key, [0, a, b])[0] += 1
    return orig_tn(a, b, out_dtype, out, splits)
ops.gemm_tn = patched_tn
nn = collections.OrderedDict()
orig_nn = ops.gemm_nn
def patched_nn(a, w, splits=None):
    key = (a.shape[0], a.shape[1], w.shape[1])
    nn.setdefault(key, [0, a, w])[0] += 1
    return orig_nn(a, w, splits)
ops.gemm_nn = patched_nn
import lbm_amd.autograd as AG
keep = pipe.fit_step(batch)
torch.cuda.synchronize()
ops._igemm = orig; ops.gemm_tn = orig_tn; ops.gemm_nn = orig_nn
stream = ops._stream()
rows = []
for key, (d, count) in seen.items():
    best = 1e12
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            _C.lib.lbmk_igemm(ctypes.byref(d), stream)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 3 * 1e3)
    M, N, K, KH, s, up, z, act, f32, gn, cin = key
    rows.append({"kind": "igemm", "M": M, "N": N, "K": K, "KH": KH, "stride": s, "up2": up, "batch": z, "act": act, "f32": f32, "gn": gn,
                 "count": count, "us": round(best, 1), "ms_total": round(best * count / 1e3, 2), "tflops": round(2.0 * M * N * K * z / best / 1e6)})
for (M, N, K), (count, a, b) in tn.items():
    best = 1e12
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            orig_tn(a, b)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3 * 1e3)
    rows.append({"kind": "gemm_tn", "M": M, "N": N, "K": K, "count": count, "us": round(best, 1), "ms_total": round(best * count / 1e3, 2),
                 "tflops": round(2.0 * M * N * K / best / 1e6)})
for (M, N, K), (count, a, w) in nn.items():
    best = 1e12
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            orig_nn(a, w)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3 * 1e3)
    rows.append({"kind": "gemm_nn", "M": M, "N": N, "K": K, "count": count, "us": round(best, 1), "ms_total": round(best * count / 1e3, 2),
                 "tflops": round(2.0 * M * N * K / best / 1e6)})
rows.sort(key=lambda r: -r["ms_total"])
print(json.dumps({"total_ms": round(sum(r["ms_total"] for r in rows), 1), "shapes": len(rows)}))
for r in rows:
    print(json.dumps(r))
