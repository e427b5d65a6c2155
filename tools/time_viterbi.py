#!/usr/bin/env python3
"""Developer probe: GPU-bound duration of the best-path alignment kernel at cfg 3 (and a batch sweep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, N, L = 400, 40, 30
dev = "cuda:0"
for B in (64, 512, 4096):
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    be = torch_asg_amd.asg.native()
    for _ in range(3): be.viterbi(x, tg, tr, il, tl)
    torch.cuda.synchronize()
    K = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000); e0.record()
    for _ in range(K): be.viterbi(x, tg, tr, il, tl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    abytes = T * B * L * 4 + B * T * 8 + B * L * 8
    print("B=%5d  %7.1f us  %9.0f utt/s  algorithmic %.1f GB/s" % (B, us, B / us * 1e6, abytes / us / 1e3))
