#!/usr/bin/env python3
"""Developer check (GPU): the sixteen-utterances-per-wavefront forward route (B >= 2048) against the same utterances run in
chunks of 64 through the fused route; then step time at B = 4096."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch_asg_amd
import util
dev = "cuda:0"

def run(x, tg, tr, il, tl, chunk=None):
    N = tr.shape[0]
    m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    B = x.shape[1]
    if chunk is None:
        xd = x.to(dev).requires_grad_(True)
        loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev)); loss.sum().backward()
        return loss.detach().cpu(), xd.grad.cpu(), m.transition.grad.cpu()
    ls, gs = [], []
    for s in range(0, B, chunk):
        xd = x[:, s:s + chunk].contiguous().to(dev).requires_grad_(True)
        loss = m(xd, tg[s:s + chunk].to(dev), il[s:s + chunk].to(dev), tl[s:s + chunk].to(dev)); loss.sum().backward()
        ls.append(loss.detach().cpu()); gs.append(xd.grad.cpu())
    return torch.cat(ls), torch.cat(gs, 1), m.transition.grad.cpu()

ok = True
for (T, B, N, L) in [(37, 2100, 40, 10), (20, 2050, 13, 5), (50, 2048, 64, 20), (33, 2049, 5, 3), (9, 2048, 33, 4), (401, 2048, 40, 30)]:
    tr, x, tg, il, tl = util.synth(T, B, N, L, T + N, True)
    a = run(x, tg, tr, il, tl)
    b = run(x, tg, tr, il, tl, 64)
    for name, u, v in zip(("loss", "grad_inputs", "grad_transition"), a, b):
        err = float((u - v).abs().max()) / max(1.0, float(v.abs().max()))
        bad = not (err < 1e-4) or not bool(torch.isfinite(u).all())
        ok &= not bad
        print("T%d B%d N%d L%d %-16s rel err %.2e %s" % (T, B, N, L, name, err, "FAIL" if bad else ""), flush=True)
print("ALL OK" if ok else "FAILURES", flush=True)
T, B, N, L = 400, 4096, 40, 30
tr, x, tg, il, tl = util.synth(T, B, N, L, 1, False)
m = torch_asg_amd.ASGLoss(N).to(dev)
xd = x.to(dev).requires_grad_(True); tgd = tg.to(dev); ild = il.to(dev); tld = tl.to(dev)
for it in range(8):
    m.transition.grad = None; xd.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(xd, tgd, ild, tld).backward()
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("B=4096 eager step %.0f us" % ((t1 - t0) * 1e6))
