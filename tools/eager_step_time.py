#!/usr/bin/env python3
"""Developer probe (GPU): the EAGER training step (no hipGraph) -- what a user who swaps the import runs.

For each host route of the step
    node      forward, autograd node and backward in C++ (csrc/binding.cpp: Fast.loss_apply, AsgLossNode)   [default]
    function  Python autograd.Function around the C++ calls (ASG_NO_CPP_NODE=1; rounds 3-5)
    python    Python statements + ctypes (ASG_NO_BINDING=1; rounds 1-2)
prints  us per step at cfg 3 (T=400: the GPU needs ~63 us, so anything above that is host-bound) and at T=16 (the
kernels take a few us: the number IS the host cost of a step), for launch_mode single and streams; and, beside them,
PyTorch's own floor on this box: a do-nothing Python autograd.Function and a do-nothing C++-free `x.sum().backward()`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
import bench
from torch_asg_amd import asg as A

dev = "cuda:0"
K = int(os.environ.get("K", "600"))


def timed(step, k=K):
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / k * 1e6)
    return best


def asg_step(T, launch):
    tr, x, tg, il, tl = bench.synth(0, dev, T=T, L=min(bench.L, T // 2))
    m = torch_asg_amd.ASGLoss(bench.N, launch_mode=launch).to(dev)
    with torch.no_grad():
        m.transition.copy_(tr)
    x.requires_grad_(True)

    def step():
        m.transition.grad = None
        x.grad = None
        m(x, tg, il, tl).backward()
    return step


def floors():
    x = torch.randn(400, 64, 40, device=dev, requires_grad=True)
    tr = torch.randn(40, 40, device=dev, requires_grad=True)
    gx, gt, out = torch.zeros_like(x), torch.zeros_like(tr), torch.zeros((), device=dev)

    class Nop(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            ctx.save_for_backward(a, b)
            return out.clone()

        @staticmethod
        def backward(ctx, g):
            a, b = ctx.saved_tensors
            return gx, gt

    def nop_step():
        x.grad = None
        tr.grad = None
        Nop.apply(x, tr).backward()

    small = torch.randn(64, device=dev, requires_grad=True)

    def sum_step():
        small.grad = None
        small.sum().backward()
    print("floor: do-nothing Python autograd.Function, forward+backward  %6.1f us / step" % timed(nop_step))
    print("floor: x.sum().backward() on 64 floats (two kernels)          %6.1f us / step" % timed(sum_step))


be = A.native()
bd = be.binding
print("box: %s, %d host cpus, torch %s" % (torch.cuda.get_device_name(0), os.cpu_count(), torch.__version__))
floors()
for launch in ("single", "streams"):
    for T in (400, 16):
        row = []
        for route in ("node", "function", "python"):
            be.binding = None if route == "python" else bd
            A._CPP_NODE = route == "node"
            try:
                row.append("%s %6.1f" % (route, timed(asg_step(T, launch))))
            finally:
                be.binding = bd
                A._CPP_NODE = True
        print("launch_mode=%-7s T=%-3d B=%d N=%d  us / eager step:  %s" % (launch, T, bench.B, bench.N, "   ".join(row)))
