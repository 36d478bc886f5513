"""Probe: capture one inference step (encode -> sample -> decode) into a HIP graph and time replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
model = bench.build_model(dev, res)
images = (torch.rand(B, 3, res, res) * 2 - 1).to(torch.bfloat16).to(dev)
eps = torch.randn(B, 4, res // 8, res // 8).to(torch.bfloat16).to(dev)
for _ in range(2):
    ref = bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ref = bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    bench.one_step(model, images, eps, 1)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = bench.one_step(model, images, eps, 1)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("max |graph - eager|:", float((out.float() - ref.float()).abs().max()))
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print(f"graph: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step")
