#!/usr/bin/env python3
"""Developer check (GPU): tiny alphabets, where grad_transition is an (almost) exact cancellation of two sums of
magnitude sum(lengths) -- the cases that expose biased roundings.  Prints the plain-rule scaled errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch_asg_amd
from oracle import asg_oracle as orc
dev = "cuda:0"
worst = 0.0
for (T, B, N, L, scale, seed) in [(257, 11, 1, 38, 0.1, 0), (600, 12, 1, 20, 1.0, 1), (1500, 12, 1, 30, 5.0, 2), (1500, 12, 2, 30, 1.0, 3),
                                  (600, 12, 3, 40, 0.1, 4), (400, 64, 40, 30, 1.0, 5), (1500, 8, 1, 1, 1.0, 6)]:
    g = torch.Generator().manual_seed(seed)
    tr = (torch.rand(N, N, generator=g) - 0.5) * 8.0
    x = torch.randn(T, B, N, generator=g) * scale
    tg = torch.randint(0, N, (B, L), generator=g)
    il = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    tl = torch.minimum(torch.randint(1, L + 1, (B,), generator=g), il)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    for mode in ("single", "streams"):
        m = torch_asg_amd.ASGLoss(N, reduction="none", launch_mode=mode).to(dev)
        with torch.no_grad(): m.transition.copy_(tr)
        xd = x.to(dev).requires_grad_(True)
        loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev)); loss.sum().backward(); torch.cuda.synchronize()
        errs = []
        for k, a in (("loss", loss.detach().cpu().numpy()), ("grad_inputs", xd.grad.cpu().numpy()), ("grad_transition", m.transition.grad.cpu().numpy())):
            ref = o[k]
            errs.append(float(np.abs(a - ref).max()) / max(1.0, float(np.abs(ref).max())))
        if mode == "single": worst = max(worst, max(errs))
        print("T%d B%d N%d L%d scale %g  %-7s loss %.2e  grad_inputs %.2e  grad_transition %.2e%s" %
              (T, B, N, L, scale, mode, errs[0], errs[1], errs[2], "   <-- over 1e-4" if max(errs) > 1e-4 else ""))
sys.exit(1 if worst > 1e-4 else 0)
