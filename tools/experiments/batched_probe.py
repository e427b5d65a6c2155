#!/usr/bin/env python3
"""Developer probe (GPU, -DASG_X16_PROBE build named by ASG_HIP_LIB): cycles per step of the batched full-lattice kernel's phases
(wavefront 0 of workgroup 0 = the beta direction of group 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, N = 400, 40
dev = "cuda:0"
be = torch_asg_amd.asg.native()
for B in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev)
    for _ in range(2): scores, state = be.full_forward(x, tr, il)
    torch.cuda.synchronize()
    v = state.view(torch.int64) if state.numel() % 8 == 0 else state[:state.numel() // 8 * 8].view(torch.int64)
    idx = (v == 0x1234567890abcdef).nonzero()
    if idx.numel() == 0:
        print("B=%d: no probe record" % B); continue
    i = int(idx[-1])
    rec = v[i + 1:i + 7].tolist()
    n = max(rec[5], 1)
    print("B=%5d steps %d | per step: transposes+write %.0f, barrier %.0f, read %.0f, products+shadow %.0f, tail %.0f  (sum %.0f cycles)"
          % (B, n, rec[0] / n, rec[1] / n, rec[2] / n, rec[3] / n, rec[4] / n, sum(rec[:5]) / n))
