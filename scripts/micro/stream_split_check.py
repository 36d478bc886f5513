"""Full-size check of the two-stream batch split: the headline step (1024^2, batch 16) returns the same bits with the
split, without it, and for images 3..5 run as a batch of three."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from lbm_amd import streams
dev = torch.device("cuda:0")
model = bench.build_model(dev, 1024)
B = 16
images = (torch.rand(B, 3, 1024, 1024) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, 128, 128).to(torch.bfloat16).to(dev)
with torch.no_grad():
    a = bench.one_step(model, images, eps, 1)
    b = bench.one_step(model, images, eps, 1)
    streams._SPLIT = False
    c = bench.one_step(model, images, eps, 1)
    d = bench.one_step(model, images[3:6].contiguous(), eps[3:6].contiguous(), 1)
    streams._SPLIT = True
torch.cuda.synchronize()
print("split repeatable:", torch.equal(a, b), " split == one stream:", torch.equal(a, c), " batch of three == members 3..5:", torch.equal(d, a[3:6]))
print("peak memory GB:", round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
