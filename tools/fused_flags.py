#!/usr/bin/env python3
"""Developer probe (GPU): which utterances the fused forward flagged, and forward / backward kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch_asg_amd, bench
from torch_asg_amd import _lib
dev = "cuda:0"
T, B, N, L = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (400, 64, 40, 30))]
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
loss, saved = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
torch.cuda.synchronize()
ws, gin = saved.tensors
sc, stb, fs = saved.sizes
tiles = (B * 2 * N * N * 4 + 255) // 256 * 256
flags = ws[sc + stb + tiles: sc + stb + tiles + 4 * B].view(torch.int32).cpu().numpy()
print("mode", saved.mode, "flagged", int(flags.sum()), "of", B, "loss", float(loss))
one = torch.ones((), device=dev)
def timed(fn, K=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
print("forward %.1f us" % timed(lambda: be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)))
def fb():
    l, sv = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
    be.loss_backward(sv, sv.tensors, one, x, tg, tr, il, tl, "mean")
print("forward+backward %.1f us" % timed(fb))
if os.environ.get("ASG_DBG"):
    al = lambda v: (v + 255) // 256 * 256
    S = L
    off = 0
    for sz in (B * T * N * 4, B * T * N * 4, B * T * S * 4, B * T * S * 4, B * T * 2 * 4, N * ((N + 7) // 8 * 8) * 4, N * 4, B * S * 2 * 4, B * S * 2 * 4):      # (asg_api.hip make_layout: ah bh ab bb klog ehat rmax asu asi | dbg)
        off = al(off + sz)
    be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
    l2, sv2 = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
    torch.cuda.synchronize()
    d = sv2.tensors[0][sc + off: sc + off + 512].view(torch.int64).cpu().numpy()
    names = ["main-a", "main-b", "cons-a", "cons-b", "ali-a", "ali-b", "afin-a", "afin-b", "rowfin-a", "rowfin-b"]
    what = {"main": ["poll e 1st half", "poll e 2nd half", "wall ticks (100 MHz)"], "cons": ["slot 1st half", "other side st_done", "slot 2nd half", "row ring space"],
            "ali": ["ring space"], "afin": ["other ast_done", "ar_done", "vmcnt(20) before publish", "loads+lds landed"], "rowfin": ["(per own group, cycles: exp stage)", "half-group 0", "half-group 4", "flush"]}
    for r, nm in enumerate(names):
        v = d[r * 5: r * 5 + 5]
        w = what[nm.split("-")[0]]
        if r == 0:
            e = d[50:55]
            print("full WG phases (cycles): init %d  roles %d  wait aligned-done %d  tile+epilogue %d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]))
            print("   role end (cycles since the roles began): " + " ".join("wave%d %d" % (k, d[57 + k] - e[1]) for k in (2, 3, 4, 5, 6)))
            print("   since the roles began: aligned workgroup done at %d (other compute unit: clocks may be offset), adone seen at %d after %d polls, edge fetch done at %d" % (d[55] - e[1], d[57] - e[1], d[58], d[56] - e[1]))
        if nm.startswith("rowfin"):
            ng = max(int(v[4]), 1)
            print("cons-%s per own group (%d groups): reads->exp %d  half-group(0) %d  half-group(4) %d  flush %d" % (nm[-1], ng, v[0] // ng, v[1] // ng, v[2] // ng, v[3] // ng))
            continue
        print("%-7s total %8d cyc (%.1f us @2.4GHz)  " % (nm, v[0], v[0] / 2400.0) + "  ".join("%s %d" % (w[k], v[1 + k]) for k in range(len(w))))
