#!/usr/bin/env python3
"""Developer probe: recursion-kernel time, one workgroup per chain (flags=2) vs one workgroup per utterance (flags=2|16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (400, 64, 40, 30))]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
K = 40
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
for flags in (2, 2 | 16, 2 | 16 | 32, 2 | 16 | 64, 2, 2 | 16, 2 | 16 | 32, 2 | 16 | 64):
    full, ali, st = be.forward(x, tg, tr, il, tl, flags)
    print("flags %2d: %.1f us   scores %.5f %.5f" % (flags, timed(lambda: be.forward(x, tg, tr, il, tl, flags)), float(full.mean()), float(ali.mean())))
