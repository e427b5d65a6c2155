#!/usr/bin/env python3
"""Developer probe (GPU): step time (hipGraph replay, 5 steps per graph) of ASGLoss forward+backward for shapes given as
T,B,N,L on the command line -- e.g. the long-target shapes of letter-based speech models (S > 64 leaves the fused step).
ASG_DTYPE=f64: in double precision."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch_asg_amd, util
dev = "cuda:0"
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1000, 64, 40, 64), (1000, 64, 40, 100), (1000, 64, 40, 200)]
for T, B, N, L in shapes:
    tr, x, tg, il, tl = util.synth(T, B, N, L, 0, True)
    m = torch_asg_amd.ASGLoss(N, launch_mode=os.environ.get("ASG_MODE", "single")).to(dev)
    if os.environ.get("ASG_DTYPE") == "f64":
        m = m.double(); tr = tr.double(); x = x.double()
    with torch.no_grad(): m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True); tgd, ild, tld = tg.to(dev), il.to(dev), tl.to(dev)
    one = torch.ones((), device=dev, dtype=xd.dtype)
    def step():
        m.transition.grad = None; xd.grad = None
        m(xd, tgd, ild, tld).backward(one)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(5): step()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("T=%d B=%d N=%d L=%d: %.1f us/step = %.0f utt/s (%.0f ns per frame)" % (T, B, N, L, dt * 1e6, B / dt, dt / T * 1e9))
