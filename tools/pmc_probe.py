#!/usr/bin/env python3
"""Workload for the PMC passes: a calibration copy (known bytes) followed by cfg-3 fwd+bwd steps (eager)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
import bench

dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(1000, dev)
m = torch_asg_amd.ASGLoss(bench.N, launch_mode="single").to(dev)
with torch.no_grad():
    m.transition.copy_(tr)
x.requires_grad_(True)
# calibration: elementwise copy of a 256 MiB fp32 tensor (reads 256 MiB, writes 256 MiB; bigger than L2, = MALL size)
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)
torch.cuda.synchronize()
for _ in range(10):
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward()
torch.cuda.synchronize()
