import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model = bench.build_model(dev, 1024)
unet = model.denoiser
z = torch.randn(2, 4, 128, 128, device=dev).bfloat16()
t = torch.full((2,), 1000.0, device=dev)
runs = []
for r in range(2):
    rec = []
    hooks = []
    for name, m in unet.named_modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: rec.append((name, o.detach().clone() if torch.is_tensor(o) else None))))
    with torch.no_grad():
        unet(sample=z, timestep=t, conditioning={"cond": {}})
    for h in hooks: h.remove()
    runs.append(rec)
n = 0
for (n1, a), (n2, b) in zip(*runs):
    if a is not None and not torch.equal(a, b):
        print("first differing leaf module:", n1, tuple(a.shape), float((a.float() - b.float()).abs().max()))
        n += 1
        if n >= 4: break
print("leaves compared", len(runs[0]))
