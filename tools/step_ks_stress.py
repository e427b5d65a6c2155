#!/usr/bin/env python3
"""Developer probe (GPU): repeat one small large-alphabet problem (8 K slices per tile) many times and count runs whose outputs differ
from the first run's -- the cross-workgroup exchange of the streaming step must be deterministic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch_asg_amd, util
dev = "cuda:0"
T, B, N, L = 7, 3, 2100, 3
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev)
with torch.no_grad(): m.transition.copy_(tr)
xd = x.to(dev); tgd, ild, tld = tg.to(dev), il.to(dev), tl.to(dev)
be = torch_asg_amd.asg.native()
first = None; bad = 0
for r in range(reps):
    full, ali, st = be.forward(xd, tgd, m.transition.detach(), ild, tld, 0)
    torch.cuda.synchronize()
    cur = full.cpu().numpy().copy()
    if first is None: first = cur
    elif not np.array_equal(first, cur):
        bad += 1
        if bad <= 3: print("run", r, "differs:", cur - first)
print("%d of %d runs differ from the first" % (bad, reps))
