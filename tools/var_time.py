#!/usr/bin/env python3
"""Developer probe: time back-to-back asg_forward launches (GPU-bound) for the library in $ASG_HIP_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (400, 64, 40, 30))]
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
gf = torch.full((B,), 1.0 / B, device=dev); ga = -gf
for _ in range(5):
    full, ali, st = be.forward(x, tg, tr, il, tl, flags)
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K):
    full, ali, st = be.forward(x, tg, tr, il, tl, flags)
torch.cuda.synchronize()
tf = (time.perf_counter() - t0) / K * 1e6
t0 = time.perf_counter()
for _ in range(K):
    be.backward(st, gf, ga, x, tg, tr, il, tl)
torch.cuda.synchronize()
tb = (time.perf_counter() - t0) / K * 1e6
print("%-40s fwd %.1f us  bwd %.1f us  loss %.4f" % (os.path.basename(os.environ.get("ASG_HIP_LIB", "default")), tf, tb, float((full - ali).mean())))
