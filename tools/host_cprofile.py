#!/usr/bin/env python3
"""Developer probe (GPU): cProfile of the eager cfg-3 training step (host side): where the Python time goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
dev = "cuda:0"
tr, x, tg, il, tl = bench.synth(0, dev)
m = torch_asg_amd.ASGLoss(bench.N).to(dev)
with torch.no_grad(): m.transition.copy_(tr)
x.requires_grad_(True)
one = torch.ones((), device=dev)
def step():
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward(one)
for _ in range(200): step()
torch.cuda.synchronize()
K = 3000
pr = cProfile.Profile()
pr.enable()
for _ in range(K): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
out = s.getvalue()
# per-call microseconds
for line in out.splitlines():
    print(line[:150])
print("(divide tottime by %d calls x 1e6 for us per step)" % K)
