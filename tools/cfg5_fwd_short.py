#!/usr/bin/env python3
"""Developer probe (GPU): a short forward at cfg 5's width (T=40 B=32 N=10000; or B N from the command line) -- for probe builds that
print from inside the step kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 40, 32, 10000, 10
if len(sys.argv) > 2:
    B, N = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tr = torch.rand(N, N, generator=g, device=dev); x = torch.randn(T, B, N, generator=g, device=dev)
tg = torch.randint(0, N, (B, L), generator=g, device=dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
full, ali, st = be.forward(x, tg, tr, il, tl, 0)
torch.cuda.synchronize()
print("mean full score %.4f" % float(full.mean()))
