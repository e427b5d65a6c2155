#!/usr/bin/env python3
"""Developer probe (GPU): forward+backward of the large-alphabet path at full cfg-5 size, backward time alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 2000, 32, 10000, 60
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tr = torch.rand(N, N, generator=g, device=dev); x = torch.randn(T, B, N, generator=g, device=dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g, device=dev)
il = torch.randint(T // 2, T + 1, (B,), generator=g, device=dev); tl = torch.randint(L // 2, L + 1, (B,), generator=g, device=dev)
m = torch_asg_amd.ASGLoss(N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
for rep in range(2):
    m.transition.grad = None; x.grad = None
    loss = m(x, tg, il, tl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("backward %.1f ms; loss %.4f; |grad_transition| %.6e; |grad_inputs| %.6e" % (dt * 1e3, float(loss), float(m.transition.grad.abs().sum()), float(x.grad.abs().sum())))
