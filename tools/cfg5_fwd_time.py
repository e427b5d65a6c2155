#!/usr/bin/env python3
"""Developer probe (GPU): forward of the generic path at full cfg-5 size, per step launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 2000, 32, 10000, 60
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tr = torch.rand(N, N, generator=g, device=dev); x = torch.randn(T, B, N, generator=g, device=dev)
tg = torch.randint(0, N, (B, L), generator=g, device=dev)
il = torch.randint(T // 2, T + 1, (B,), generator=g, device=dev); tl = torch.randint(L // 2, L + 1, (B,), generator=g, device=dev)
be = torch_asg_amd.asg.native()
flags = int(os.environ.get("ASG_FWD_FLAGS", "0"))          # 0: one stream; torch_asg_amd._lib.FLAG_SINGLE_LAUNCH: what ASGLoss passes (two streams on this route)
full, ali, st = be.forward(x, tg, tr, il, tl, flags)
torch.cuda.synchronize(); t0 = time.perf_counter()
full, ali, st = be.forward(x, tg, tr, il, tl, flags)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("forward %.1f ms = %.1f us per step launch = %.2f TB/s of E/F; mean full score %.4f" % (dt * 1e3, dt / (T - 1) * 1e6, 2 * N * N * 4 / (dt / (T - 1)) / 1e12, float(full.mean())))
