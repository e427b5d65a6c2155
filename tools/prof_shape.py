#!/usr/bin/env python3
"""Developer probe (GPU, under rocprofv3 --kernel-trace): three training steps of ASGLoss at the shape T,B,N,L given on the command
line; tools/kernel_medians.py then lists the kernels."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch_asg_amd, util
T, B, N, L = [int(v) for v in sys.argv[1].split(",")]
dev = "cuda:0"
tr, x, tg, il, tl = util.synth(T, B, N, L, 0, True)
m = torch_asg_amd.ASGLoss(N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
xd = x.to(dev).requires_grad_(True)
for _ in range(3):
    m.transition.grad = None; xd.grad = None
    m(xd, tg.to(dev), il.to(dev), tl.to(dev)).backward()
torch.cuda.synchronize()
