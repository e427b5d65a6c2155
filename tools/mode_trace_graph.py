#!/usr/bin/env python3
"""Developer probe (GPU, under rocprofv3 --kernel-trace): hipGraph replay of 4 cfg-3 steps in a launch mode (argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 400, 64, 40, 30
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
m = torch_asg_amd.ASGLoss(N, launch_mode=sys.argv[1]).to(dev)
one = torch.ones((), device=dev)
def step():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward(one)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4): step()
for _ in range(3): gr.replay()
torch.cuda.synchronize()
