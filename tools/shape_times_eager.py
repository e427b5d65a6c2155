#!/usr/bin/env python3
"""Developer probe (GPU): EAGER step time (no hipGraph) of ASGLoss forward+backward for shapes T,B,N,L."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch_asg_amd, util
dev = "cuda:0"
for a in sys.argv[1:]:
    T, B, N, L = (int(v) for v in a.split(","))
    tr, x, tg, il, tl = util.synth(T, B, N, L, 0, True)
    m = torch_asg_amd.ASGLoss(N).to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True); tgd, ild, tld = tg.to(dev), il.to(dev), tl.to(dev)
    def step():
        m.transition.grad = None; xd.grad = None
        m(xd, tgd, ild, tld).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("eager T=%d B=%d N=%d L=%d: %.1f us/step (%.0f ns per frame)" % (T, B, N, L, dt * 1e6, dt / T * 1e9))
