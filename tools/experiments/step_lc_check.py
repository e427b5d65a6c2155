#!/usr/bin/env python3
"""Developer probe (GPU): the loader / consumer step kernel (default) against the one-wavefront-does-both kernel (ASG_STEP_LC=0):
forward scores, states-derived gradients bit for bit on a few shapes incl. a tail chunk (N % 32 != 0), B < 32, several row tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch_asg_amd, util
dev = "cuda:0"
def run(T, B, N, L, lc):
    os.environ["ASG_STEP_LC"] = lc
    tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
    m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True)
    loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev)); loss.sum().backward(); torch.cuda.synchronize()
    return loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), m.transition.grad.cpu().numpy()
for shape in [(9, 3, 2100, 3), (12, 34, 2500, 4), (7, 32, 4111, 2), (20, 5, 3000, 6)]:
    a = run(*shape, "1"); b = run(*shape, "0")
    print(shape, "bit-identical:", [bool(np.array_equal(p, q)) for p, q in zip(a, b)], "finite:", bool(np.isfinite(a[0]).all()))
