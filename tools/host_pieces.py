#!/usr/bin/env python3
"""Developer probe (GPU): host time of the pieces of one eager ASGLoss step (no device sync inside the loop)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
from torch_asg_amd import _lib
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(0, dev)
m = torch_asg_amd.ASGLoss(bench.N, launch_mode="single").to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
be = torch_asg_amd.asg.native()
one = torch.ones((), device=dev)
K = 300
def timeit(name, fn, sync_every=50):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = 0.0
    for k in range(K):
        t0 = time.perf_counter(); fn(); t += time.perf_counter() - t0
        if k % sync_every == sync_every - 1: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-52s %7.1f us" % (name, t / K * 1e6))
xd = x.detach()
timeit("be.loss_forward (C call + buffers)", lambda: be.loss_forward(xd, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH))
l, sv = be.loss_forward(xd, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
timeit("be.loss_backward", lambda: be.loss_backward(sv, sv.tensors, one, xd, tg, tr, il, tl, "mean"))
timeit("module forward (autograd Function.apply)", lambda: m(x, tg, il, tl))
def fb():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward(one)
timeit("module forward + backward", fb)
p, keep = be._problem(xd, tr, tg, il, tl)
timeit("  _problem", lambda: be._problem(xd, tr, tg, il, tl))
timeit("  _check", lambda: be._check(xd, tr, tg, il, tl))
timeit("  torch.empty x3", lambda: (torch.empty((), device=dev), torch.empty(30000000, dtype=torch.uint8, device=dev), torch.empty(400, 64, 40, device=dev)))
timeit("  current_stream", lambda: torch.cuda.current_stream(0).cuda_stream)
