#!/usr/bin/env python3
"""Developer probe: time the generic (large-alphabet) path, check size-independent properties."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2000, 32, 10000, 60))]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tr = torch.rand(N, N, generator=g, device=dev)
x = torch.randn(T, B, N, generator=g, device=dev)
tg = torch.randint(0, N, (B, L), generator=g, device=dev)
il = torch.randint(T // 2, T + 1, (B,), generator=g, device=dev)
tl = torch.randint(max(1, L // 2), L + 1, (B,), generator=g, device=dev)
be = torch_asg_amd.asg.native()
from torch_asg_amd import _lib
torch.cuda.synchronize(); t0 = time.perf_counter()
full, ali, st = be.forward(x, tg, tr, il, tl, _lib.FLAG_ALPHA_SCORES)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("forward %.1f ms  state %.2f GB" % ((t1 - t0) * 1e3, st.numel() / 1e9))
fa, fb = full[B:], full[:B]
print(fa[:3].tolist(), fb[:3].tolist())
print("alpha/beta full score rel diff", float(((fa - fb).abs() / fb.abs()).max()), " aligned", float(((ali[B:] - ali[:B]).abs() / ali[:B].abs()).max()))
gf = torch.full((B,), 1.0 / B, device=dev)
t0 = time.perf_counter()
gtr, gin = be.backward(st, gf, -gf, x, tg, tr, il, tl)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("backward %.1f ms" % ((t1 - t0) * 1e3))
print("finite:", bool(torch.isfinite(gtr).all()), bool(torch.isfinite(gin).all()))
rows = gin.sum(-1)            # [T,B]: full posterior (sums to g) minus aligned posterior (sums to g) = 0 on valid frames
valid = torch.arange(T, device=dev)[:, None] < il[None, :]
print("max |sum_i grad_inputs| valid frames %.3e ; padded frames max |g| %.3e" % (float(rows[valid].abs().max()), float(gin[~valid].abs().max()) if (~valid).any() else 0.0))
print("sum grad_transition %.3e (abs sum %.3e)" % (float(gtr.double().sum()), float(gtr.double().abs().sum())))
