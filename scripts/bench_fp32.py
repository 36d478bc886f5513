"""fp32 mode, 512x512, batch 1, 1 NFE (the reference's CPU-runnable configuration) on one GPU: python scripts/bench_fp32.py [--steps N]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--batch", type=int, default=1)
args = ap.parse_args()
device = torch.device("cuda:0")
torch.cuda.set_device(device)
print(json.dumps(bench.extra_fp32(device, 0, 1, None, args.res, args.batch, args.steps)))
