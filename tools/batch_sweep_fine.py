#!/usr/bin/env python3
"""Developer probe: fwd+bwd GPU time vs batch size at T=400 N=40 L=30, default launch mode, fine grid (graph replay of 10 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, N, L = 400, 40, 30
dev = "cuda:0"
for B in [int(a) for a in sys.argv[1:]] or (64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 768, 1024, 1536, 2048, 3072, 4096):
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
    for _ in range(10): gr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    abytes = 2 * T * B * N * 4 + 2 * N * N * 4 + B * (8 * L + 20)
    print("B=%5d  %8.1f us/step  %9.0f utt/s  algorithmic %.1f GB/s (%.2f%% of 8 TB/s)" % (B, dt * 1e6, B / dt, abytes / dt / 1e9, abytes / dt / 8e12 * 100))
