#!/usr/bin/env python3
"""Workload for the PMC passes of the stand-alone route: calibration copy, then 3 eager training steps at T=400 B=512 N=40 L=30."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 400, 512, 40, 30
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)
torch.cuda.synchronize()
m = torch_asg_amd.ASGLoss(N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
for _ in range(3):
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward()
torch.cuda.synchronize()
