#!/usr/bin/env python3
"""Developer probe (GPU): the shared-matrix step (ASG_STEP2=1) against the two-copy step and the fp64 oracle on small T."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch, torch_asg_amd, util
from oracle import asg_oracle as orc
dev = "cuda:0"
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(6, 3, 4100, 3), (5, 2, 6200, 3), (4, 40, 4096, 2)]
for T, B, N, L in shapes:
    tr, x, tg, il, tl = util.synth(T, B, N, L, 3, True)
    res = {}
    for env in ("0", "1"):
        os.environ["ASG_STEP2"] = env
        m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev)
        with torch.no_grad(): m.transition.copy_(tr)
        xd = x.to(dev).requires_grad_(True)
        t0 = time.time()
        loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev))
        loss.sum().backward(); torch.cuda.synchronize()
        res[env] = (loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), m.transition.grad.cpu().numpy(), time.time() - t0)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    for env in ("0", "1"):
        errs = [util.tol_ok(res[env][k], o[key], 1e-4)[1] for k, key in enumerate(("loss", "grad_inputs", "grad_transition"))]
        print("T%d B%d N%d L%d STEP2=%s: scaled errors vs oracle loss %.2e grad_inputs %.2e grad_transition %.2e  (%.0f ms)" % ((T, B, N, L, env) + tuple(errs) + (res[env][3] * 1e3,)))
    print("   two routes differ by", [float(np.nanmax(np.abs(res["0"][k] - res["1"][k]))) for k in range(3)])
