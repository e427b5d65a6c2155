#!/usr/bin/env python3
"""Workload for the PMC passes of the large-alphabet step: a calibration copy (known bytes), then ONE forward + backward at
T=60 B=32 N=10000 L=20 (59 launches of fwd_step_kernel, each streaming the same matrices as at cfg 5's T=2000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
dev = "cuda:0"
T, B, N, L = 60, 32, 10000, 20
g = torch.Generator(device=dev).manual_seed(0)
tr = torch.rand(N, N, generator=g, device=dev); x = torch.randn(T, B, N, generator=g, device=dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g, device=dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)
torch.cuda.synchronize()
m = torch_asg_amd.ASGLoss(N).to(dev)
with torch.no_grad():
    m.transition.copy_(tr)
m(x, tg, il, tl).backward()
torch.cuda.synchronize()
