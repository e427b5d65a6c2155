#!/usr/bin/env python3
"""Developer probe (GPU, under rocprofv3): FCC-only and FAC-only backward at B = 512 / 4096 (which part costs what)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
from torch_asg_amd.asg import FCC, FAC
T, N, L = 400, 40, 30
dev = "cuda:0"
for B in (512, 4096):
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev).requires_grad_(True); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    for fn in (FCC, FAC):
        for _ in range(3):
            tr.grad = None; x.grad = None
            fn.apply(tr, x, tg, il, tl).sum().backward()
        torch.cuda.synchronize()
