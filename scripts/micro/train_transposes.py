"""Which call sites transpose what in one training step (diagnostic)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from lbm_amd import ops
from lbm_amd.trainer import TrainingConfig, TrainingPipeline
dev = torch.device("cuda:0"); BF = torch.bfloat16
model = bench.build_model(dev, 512)
model.train(); model.vae.freeze()
pipe = TrainingPipeline(model, TrainingConfig(learning_rate=4e-5, optimizer_name="AdamW", trainable_params=["denoiser.*"]))
pipe.configure_optimizers()
g = torch.Generator().manual_seed(1)
batch = {"source_image": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(dev),
         "source_image_paste": (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(BF).to(dev)}
pipe.fit_step(batch); pipe.fit_step(batch)
torch.cuda.synchronize()
sites = collections.Counter(); byts = collections.Counter()
orig = ops.transpose
def patched(x, *a, **k):
    st = traceback.extract_stack(limit=4)
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[:-1]))
    sites[key] += 1; byts[key] += x.numel() * 2
    return orig(x, *a, **k)
ops.transpose = patched
import lbm_amd.autograd as AG, lbm_amd.models.layers as L
pipe.fit_step(batch)
torch.cuda.synchronize()
for k, v in sites.most_common(12):
    print(v, f"{byts[k]/1e6:.0f} MB", k)
