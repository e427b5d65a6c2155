#!/usr/bin/env python3
"""Developer probe (GPU): host cost of forward+backward through a do-nothing custom autograd.Function with the same
signature shape as ASGLossFunction (two differentiable inputs, saved tensors) -- PyTorch's own floor for the eager path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
x = torch.randn(400, 64, 40, device=dev, requires_grad=True)
tr = torch.randn(40, 40, device=dev, requires_grad=True)
gx = torch.zeros_like(x); gt = torch.zeros_like(tr); out = torch.zeros((), device=dev)
class Nop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return out.clone()
    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return gx, gt
one = torch.ones((), device=dev)
def step():
    x.grad = None; tr.grad = None
    Nop.apply(x, tr).backward(one)
for _ in range(50): step()
torch.cuda.synchronize()
K = 500; t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
print("no-op Function forward+backward: %.1f us per step (host)" % ((time.perf_counter() - t0) / K * 1e6))
