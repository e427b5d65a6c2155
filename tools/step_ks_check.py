#!/usr/bin/env python3
"""Developer probe (GPU, ASG_DEV_PROBES build): the fp32 streaming step with a forced number of K slices (ASG_STEP_KS) against the
fp64 oracle on one small shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch_asg_amd, util
from oracle import asg_oracle as orc
dev = "cuda:0"
T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (7, 3, 2100, 3))]
rng = np.random.default_rng(N)
tr, x, tg, _, _ = util.synth(T, B, N, L, N)
il = rng.integers(max(1, T // 2), T + 1, B); tl = np.minimum(rng.integers(1, L + 1, B), il)
o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
for ks in ("1", "2", "3", "4", "5", "6", "7", "8"):
    os.environ["ASG_STEP_KS"] = ks
    m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True)
    loss = m(xd, tg.to(dev), torch.from_numpy(il).to(dev), torch.from_numpy(tl).to(dev)); loss.sum().backward(); torch.cuda.synchronize()
    print("ks", ks, "loss", loss.detach().cpu().numpy(), "oracle", o["loss"], "err gin %.2e gtr %.2e" % (
        util.tol_ok(xd.grad.cpu().numpy(), o["grad_inputs"])[1], util.tol_ok(m.transition.grad.cpu().numpy(), o["grad_transition"])[1]))
