#!/usr/bin/env python3
"""Developer timing probe (GPU): fwd / bwd / fwd+bwd of the cfg-3 workload per launch mode and mat-vec variant."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_asg_amd
from torch_asg_amd import _lib

def synth(T, B, N, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    tr = torch.rand(N, N, generator=g); x = torch.randn(T, B, N, generator=g); tg = torch.randint(0, N, (B, L), generator=g)
    return tr, x, tg

def main():
    T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (400, 64, 40, 30))]
    dev = "cuda:0"
    tr, x, tg = synth(T, B, N, L)
    tr, x, tg = tr.to(dev), x.to(dev), tg.to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    be = torch_asg_amd.asg.native()
    gf = torch.full((B,), 1.0 / B, device=dev); ga = -gf
    def ev():
        return torch.cuda.Event(enable_timing=True)
    res = {}
    for name, flags in (("streams", 1), ("single", 2), ("serial", 0), ("single+readlane", 2 | 4), ("streams+readlane", 1 | 4)):
        for _ in range(5):
            full, ali, st = be.forward(x, tg, tr, il, tl, flags)
            be.backward(st, gf, ga, x, tg, tr, il, tl)
        torch.cuda.synchronize()
        K = 50
        e0, e1, e2 = ev(), ev(), ev()
        tf = tb = 0.0
        for _ in range(K):
            e0.record(); full, ali, st = be.forward(x, tg, tr, il, tl, flags); e1.record()
            be.backward(st, gf, ga, x, tg, tr, il, tl); e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        # back-to-back throughput (host overlap allowed)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            full, ali, st = be.forward(x, tg, tr, il, tl, flags)
            be.backward(st, gf, ga, x, tg, tr, il, tl)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res[name] = dict(fwd_us=tf / K * 1e3, bwd_us=tb / K * 1e3, loop_us=(t1 - t0) / K * 1e6)
        print(name, json.dumps(res[name]))
    # forward-only
    for _ in range(3): be.forward_only(x, tg, tr, il, tl, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): be.forward_only(x, tg, tr, il, tl, 1)
    torch.cuda.synchronize(); print("forward_only loop_us", (time.perf_counter() - t0) / 50 * 1e6)
    # whole module, eager autograd
    m = torch_asg_amd.ASGLoss(N).to(dev)
    with torch.no_grad(): m.transition.copy_(tr)
    xr = x.clone().requires_grad_(True)
    for _ in range(5):
        m.zero_grad(set_to_none=True); xr.grad = None
        m(xr, tg, il, tl).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        m.zero_grad(set_to_none=True); xr.grad = None
        m(xr, tg, il, tl).backward()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    print("ASGLoss eager fwd+bwd us", t * 1e6, "utt/s", B / t)

if __name__ == "__main__":
    main()
