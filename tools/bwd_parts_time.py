#!/usr/bin/env python3
"""Developer probe (GPU): time of the stand-alone backward (FCC-only, FAC-only, whole criterion) at a few batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
from torch_asg_amd.asg import FCC, FAC
T, N, L = 400, 40, 30
dev = "cuda:0"
Bs = [int(a) for a in sys.argv[1:]] or [512, 4096]
for B in Bs:
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev).requires_grad_(True); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    m = torch_asg_amd.ASGLoss(N).to(dev)
    out = []
    for name, fn in (("FCC", lambda: FCC.apply(tr, x, tg, il, tl).sum()), ("FAC", lambda: FAC.apply(tr, x, tg, il, tl).sum()),
                     ("ASG", lambda: m(x, tg, il, tl))):
        best = 1e9
        for _ in range(6):
            tr.grad = None; x.grad = None; m.transition.grad = None
            y = fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); y.backward(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        out.append("%s bwd %.1f us" % (name, best))
    print("B=%d  " % B + "  ".join(out), " lib", os.path.basename(os.environ.get("ASG_HIP_LIB", "default")), flush=True)
