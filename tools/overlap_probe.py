#!/usr/bin/env python3
"""Developer probe (GPU): what overlapping the gradient assembly with the recursions could buy at large batches.

Round-6 question: at B >= 96 the step is `recursion launch, then assembly launch`; the design that
could shorten it runs the assembly of the frames both directions have passed WHILE the chains still run.  Before
building the in-launch progress words, the best case is measurable with what exists: launch the recursion kernels of
step k+1 on one HIP stream and the assembly kernels of step k (its state buffer is complete) on another, at the same
moment.  No kernel waits for anything, so this is an upper bound on the overlap: the two kinds of work share the
compute units exactly as they would in the fused arrangement, minus every synchronisation cost.

Prints, per batch size: recursion launch alone, assembly launches alone, one after the other on one stream, and the
two side by side on two streams (events on both streams; eager launches through the C++ host path, so the host is
not the bound at these sizes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
from torch_asg_amd import asg as A

dev = torch.device("cuda:0")
T, N, L = 400, 40, 30
K = int(os.environ.get("K", "30"))
be = A.native()
bd = be.binding
assert bd is not None


def make(B):
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev)
    x = torch.randn(T, B, N, generator=g).to(dev)
    tg = torch.randint(0, N, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev)
    tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    return x, tr, tg, il, tl


def timed(fn, streams, k=K):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for _ in range(k):
            fn()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / k * 1e3)
    return best


one = torch.ones((), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)
print("box: %s; T=%d N=%d L=%d fp32, stand-alone route (B > 80), us per step" % (torch.cuda.get_device_name(0), T, N, L))
for B in (128, 256, 512, 1024, 4096):
    x, tr, tg, il, tl = make(B)
    args = (x, tr, tg, il, tl)

    def fwd():
        return bd.try_loss_forward(*args, 2, 2)

    r = fwd()
    assert r is not None and r[1] == 0
    rec = (0, 0, r[5], 0, 2)
    state = r[2]

    def bwd():
        return bd.try_loss_backward(rec, state, None, one, *args)

    t_f = timed(fwd, [])
    t_b = timed(bwd, [])
    t_seq = timed(lambda: (fwd(), bwd()), [])

    def side_by_side(sa, sb):
        def f():
            with torch.cuda.stream(sa):
                fwd()
            with torch.cuda.stream(sb):
                bwd()
            # the next pair starts when both are done (what a step boundary is)
            sa.wait_stream(sb)
            sb.wait_stream(sa)
        return f

    t_par = timed(side_by_side(s1, s2), [s1, s2])
    t_par_hi = timed(side_by_side(hi, s2), [hi, s2])
    print("B=%-5d recursions %7.1f   assembly %6.1f   one after the other %7.1f   side by side %7.1f   (recursions on a "
          "high-priority stream %7.1f)   best-case gain %4.1f %%" % (B, t_f, t_b, t_seq, t_par, t_par_hi,
                                                                    100.0 * (1.0 - min(t_par, t_par_hi) / t_seq)))
    del x, state, r
    torch.cuda.empty_cache()
