#!/usr/bin/env python3
"""Developer probe: GPU-bound duration of the forward launch (whole and per recursion chain) and of the backward.
Launches are queued behind a spin kernel so that host overhead is hidden; $ASG_HIP_LIB selects the library."""
import os, sys
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
out = []
for name, mask in (("all", 15), ("full_alpha", 1), ("full_beta", 2), ("ali_alpha", 4), ("ali_beta", 8), ("full", 3), ("ali", 12)):
    os.environ["ASG_DEBUG_MASK"] = str(mask)
    out.append("%s %.1f" % (name, timed(lambda: be.forward(x, tg, tr, il, tl, flags))))
os.environ["ASG_DEBUG_MASK"] = "15"
full, ali, st = be.forward(x, tg, tr, il, tl, flags)
out.append("bwd %.1f" % timed(lambda: be.backward(st, gf, ga, x, tg, tr, il, tl)))
print("%-28s" % os.path.basename(os.environ.get("ASG_HIP_LIB", "default")), " | ".join(out), "| loss %.4f" % float((full - ali).mean()))
