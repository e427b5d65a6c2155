#!/usr/bin/env python3
"""Developer probe: GPU-bound timing of asg_loss_backward (assembly + reduce) via back-to-back launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(0, dev)
be = torch_asg_amd.asg.native()
loss, st = be.loss_forward(x, tg, tr, il, tl, "mean", 2)
g = torch.ones((), device=dev)
for _ in range(5): be.loss_backward(st, g, x, tg, tr, il, tl, "mean")
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    be.loss_backward(st, g, x, tg, tr, il, tl, "mean")
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for _ in range(10): be.loss_backward(st, g, x, tg, tr, il, tl, "mean")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): graph.replay()
torch.cuda.synchronize()
print("ASG_BWD_WGS=%s backward (assembly+reduce) %.2f us" % (os.environ.get("ASG_BWD_WGS", "default"), (time.perf_counter() - t0) / 200 * 1e6))
