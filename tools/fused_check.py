#!/usr/bin/env python3
"""Developer check (GPU): the fused training step against the fp64 oracle on a few shapes, and its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch_asg_amd, util
from oracle import asg_oracle as orc
dev = "cuda:0"

def run(T, B, N, L, seed, variable, red, mode="single"):
    tr, x, tg, il, tl = util.synth(T, B, N, L, seed, variable)
    tl = torch.minimum(tl, il)
    m = torch_asg_amd.ASGLoss(N, reduction=red, launch_mode=mode).to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True)
    loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev))
    w = torch.linspace(0.5, 1.5, loss.numel(), device=dev).reshape(loss.shape) if red == "none" else torch.ones((), device=dev)
    (loss * w).sum().backward()
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red,
                     grad_out=(w.cpu().numpy().astype(np.float64) if red == "none" else None))
    out = {}
    for k, a in (("loss", loss.detach().cpu().numpy()), ("grad_inputs", xd.grad.cpu().numpy()), ("grad_transition", m.transition.grad.cpu().numpy())):
        ok, e = util.tol_ok(a, o[k], 1e-4)
        out[k] = e
    return out

cases = [(400, 64, 40, 30, 0, False, "mean"), (400, 64, 40, 30, 0, True, "mean"), (150, 16, 30, 20, 0, True, "sum"),
         (60, 5, 21, 9, 3, True, "none"), (23, 3, 8, 7, 1, True, "sum"), (37, 4, 63, 30, 2, True, "mean"), (5, 2, 7, 3, 4, True, "none"),
         (100, 3, 17, 64, 5, True, "mean"), (16, 2, 40, 5, 6, False, "sum"), (17, 2, 40, 5, 6, False, "sum"), (33, 2, 12, 5, 6, True, "sum")]
if len(sys.argv) > 1 and sys.argv[1] == "time":
    cases = []
bad = 0
for c in cases:
    try:
        r = run(*c)
    except Exception as e:
        print(c, "EXC", e); bad += 1; continue
    flag = "" if all(v <= 1e-4 for v in r.values()) else "   <-- FAIL"
    bad += bool(flag)
    print(c, " ".join("%s %.2e" % kv for kv in r.items()), flag)
# timing of one step (graph replay), cfg 3
import bench
tr, x, tg, il, tl = bench.synth(1000, dev)
m = torch_asg_amd.ASGLoss(bench.N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
one = torch.ones((), device=dev)
def step():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward(one)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): step()
for _ in range(5): gr.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 200
for _ in range(K): gr.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print("cfg3 graph step: %.1f us  (%.0f utt/s)" % (dt * 1e6, bench.B / dt))
sys.exit(1 if bad else 0)
