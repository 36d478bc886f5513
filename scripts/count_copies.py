"""Where do non-trivial .contiguous() / .to() / .float() copies of one training step come from?  (call-site histogram)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lbm_amd.trainer import TrainingConfig, TrainingPipeline
dev = torch.device("cuda:0")
model = bench.build_model(dev, 512); model.train(); model.vae.freeze()
pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
pipe.configure_optimizers()
g = torch.Generator().manual_seed(1)
batch = {k: (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).bfloat16().to(dev) for k in ("source_image", "source_image_paste")}
pipe.fit_step(batch)
cnt = collections.Counter()
def site():
    for f in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "lbm_amd" in f.filename:
            return f"{os.path.relpath(f.filename)}:{f.lineno} {f.line.strip()[:90]}"
    return "other"
oc, ot, of_, orr = torch.Tensor.contiguous, torch.Tensor.to, torch.Tensor.float, torch.Tensor.reshape
def c(self, *a, **k):
    if not self.is_contiguous(): cnt[("contiguous", site())] += 1
    return oc(self, *a, **k)
def t(self, *a, **k):
    r = ot(self, *a, **k)
    if r.data_ptr() != self.data_ptr(): cnt[("to", site())] += 1
    return r
def fl(self, *a, **k):
    if self.dtype != torch.float32: cnt[("float", site())] += 1
    return of_(self, *a, **k)
def rs(self, *a, **k):
    r = orr(self, *a, **k)
    if r.data_ptr() != self.data_ptr() and self.numel() > 0: cnt[("reshape-copy", site())] += 1
    return r
torch.Tensor.contiguous, torch.Tensor.to, torch.Tensor.float, torch.Tensor.reshape = c, t, fl, rs
pipe.fit_step(batch)
torch.cuda.synchronize()
for (kind, where), n in cnt.most_common(30):
    print(f"{n:6d} {kind:13s} {where}")
