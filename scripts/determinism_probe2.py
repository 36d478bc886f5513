import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(4, 64, 64, 640, device=dev).bfloat16()
g = torch.ones(640, device=dev); b = torch.zeros(640, device=dev)
ys = [ops.groupnorm(x, g, b, 32, 1e-5, True) for _ in range(5)]
print("groupnorm:", all(torch.equal(ys[0], y) for y in ys))
ys = [ops.layernorm(x.view(-1, 640), g, b) for _ in range(5)]
print("layernorm:", all(torch.equal(ys[0], y) for y in ys))
q = torch.randn(4, 4096, 640, device=dev).bfloat16(); k = torch.randn(4, 4096, 640, device=dev).bfloat16(); vt = torch.randn(4, 640, 4096, device=dev).bfloat16()
ys = [ops.flash_attn(q, k, vt, 10, 64) for _ in range(5)]
print("flash:", all(torch.equal(ys[0], y) for y in ys))
w = torch.randn(1280, 640, device=dev).bfloat16() * 0.04
pw = ops.pack_linear_weight(w, torch.zeros(1280, device=dev))
for t in (-1, 0, 1, 7, 8, 10):
    ys = [ops.linear(x.view(-1, 640), pw, tile=t) for _ in range(5)]
    print("linear tile", t, all(torch.equal(ys[0], y) for y in ys))
wc = torch.randn(640, 640, 3, 3, device=dev).bfloat16() * 0.01
pc = ops.pack_conv_weight(wc, torch.zeros(640, device=dev))
ys = [ops.conv2d(x, pc) for _ in range(5)]
print("conv:", all(torch.equal(ys[0], y) for y in ys))
pg = ops.pack_geglu_weight(torch.randn(5120, 640, device=dev).bfloat16() * 0.04, torch.zeros(5120, device=dev))
ys = [ops.linear(x.view(-1, 640), pg) for _ in range(5)]
print("geglu:", all(torch.equal(ys[0], y) for y in ys))
