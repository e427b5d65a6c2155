#!/usr/bin/env python3
"""Developer probe: where does the host time of one eager ASGLoss step go?"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(0, dev)
m = torch_asg_amd.ASGLoss(bench.N, launch_mode="single").to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
def step():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("eager us/step", (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
