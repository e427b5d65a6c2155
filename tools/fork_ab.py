#!/usr/bin/env python3
"""Developer probe (GPU): graph-replay step time with and without the two-stream fork under capture (ASG_FORK_IN_CAPTURE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
dev = "cuda:0"
cases = [("cfg3 streams", 400, 64, 40, 30, "streams"), ("cfg3 serial", 400, 64, 40, 30, "serial"), ("long targets single", 1000, 64, 40, 200, "single"),
         ("N=128 single", 400, 64, 128, 30, "single"), ("N=512 single", 400, 64, 512, 30, "single"), ("B=512 streams", 400, 512, 40, 30, "streams")]
for name, T, B, N, L, mode in cases:
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    out = []
    for fk in ("1", "0"):
        os.environ["ASG_FORK_IN_CAPTURE"] = fk
        m = torch_asg_amd.ASGLoss(N, launch_mode=mode).to(dev)
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
                for _ in range(5): step()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 50 * 1e6)
    print("%-22s fork under capture %8.1f us/step   one stream under capture %8.1f us/step" % (name, out[0], out[1]))
