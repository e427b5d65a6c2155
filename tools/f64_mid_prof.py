#!/usr/bin/env python3
"""Developer probe (GPU, under rocprofv3 --kernel-trace): fp64 training steps at a medium alphabet (T=400 B=64 N=128 L=30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, B, L = 400, 64, 30
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g, dtype=torch.float64).to(dev); x = torch.randn(T, B, N, generator=g, dtype=torch.float64).to(dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g).to(dev); il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
m = torch_asg_amd.ASGLoss(N).to(dev).double()
with torch.no_grad(): m.transition.copy_(tr)
for _ in range(3):
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward()
torch.cuda.synchronize()
