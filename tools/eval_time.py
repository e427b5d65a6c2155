#!/usr/bin/env python3
"""Developer probe (GPU): forward-only (eval) loss time at cfg 3 and at a few batch sizes, graph replay of 10 calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, N, L = 400, 40, 30
dev = "cuda:0"
for B in (64, 128, 512):
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    m = torch_asg_amd.ASGLoss(N).to(dev).eval()
    with torch.no_grad():
        m.transition.copy_(tr)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): m(x, tg, il, tl)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10): out = m(x, tg, il, tl)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): gr.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("eval B=%4d  %7.1f us/call  %9.0f utt/s  loss %.4f" % (B, dt * 1e6, B / dt, float(out)), flush=True)
