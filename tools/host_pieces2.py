#!/usr/bin/env python3
"""Developer probe (GPU): host time inside ASGLossFunction.forward / .backward (the latter runs on the autograd thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
from torch_asg_amd import asg as A
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(0, dev)
m = torch_asg_amd.ASGLoss(bench.N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
acc = {"fwd": 0.0, "bwd": 0.0, "n": 0}
F = A.ASGLossFunction
of, ob = F.forward, F.backward
def tf(ctx, *a):
    t0 = time.perf_counter(); r = of(ctx, *a); acc["fwd"] += time.perf_counter() - t0; return r
def tb(ctx, g):
    t0 = time.perf_counter(); r = ob(ctx, g); acc["bwd"] += time.perf_counter() - t0; acc["n"] += 1; return r
F.forward = staticmethod(tf); F.backward = staticmethod(tb)
one = torch.ones((), device=dev)
def step():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward(one)
for _ in range(50): step()
torch.cuda.synchronize(); acc.update(fwd=0.0, bwd=0.0, n=0)
K = 400; t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / K * 1e6
print("step %.1f us host; inside Function.forward %.1f us, inside Function.backward %.1f us" % (tot, acc["fwd"] / K * 1e6, acc["bwd"] / K * 1e6))
