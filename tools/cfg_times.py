#!/usr/bin/env python3
"""Developer probe (GPU): training-step time of BASELINE.json's small-alphabet configs (graph replay of 10 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
dev = "cuda:0"
for name, (T, B, N, L) in (("cfg1", (6, 2, 7, 5)), ("cfg2", (150, 16, 30, 20)), ("cfg3", (400, 64, 40, 30)), ("cfg4 on one GPU", (400, 512, 40, 30))):
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    m = torch_asg_amd.ASGLoss(N).to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    one = torch.ones((), device=dev)
    def step():
        m.transition.grad = None; x.grad = None
        m(x, tg, il, tl).backward(one)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10): step()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%-16s T=%d B=%d N=%d L=%d  %7.1f us/step  %9.0f utt/s" % (name, T, B, N, L, dt * 1e6, B / dt), flush=True)
