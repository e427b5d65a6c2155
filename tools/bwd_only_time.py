#!/usr/bin/env python3
"""Developer probe (GPU): time of the fused backward launch alone (graph of 20 consecutive launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
from torch_asg_amd import _lib
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(1000, dev)
be = torch_asg_amd.asg.native()
one = torch.ones((), device=dev)
loss, sv = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
def bw(): be.loss_backward(sv, sv.tensors, one, x, tg, tr, il, tl, "mean")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): bw()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): bw()
for _ in range(3): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("backward launch: %.2f us" % (e0.elapsed_time(e1) / 20 * 1e3))
