#!/usr/bin/env python3
"""Developer probe (GPU): cfg-3 step time (10 steps per graph), 3 repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(1000, dev)
m = torch_asg_amd.ASGLoss(bench.N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
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
        for _ in range(10): step()
res = []
for rep in range(3):
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 200 * 1e6)
print("cfg3 step: " + " ".join("%.2f" % r for r in res) + " us")
