#!/usr/bin/env python3
"""Developer probe (GPU): per-call time of the fused forward; prints slow calls with their flagged count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch_asg_amd
from torch_asg_amd import _lib
dev = "cuda:0"
T, B, N, L = 400, 64, 40, 30
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
tiles = (B * 2 * N * N * 4 + 255) // 256 * 256
ts = []
for it in range(300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss, saved = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3
    ws, gin = saved.tensors; sc, stb, fs = saved.sizes
    flags = ws[sc + stb + tiles: sc + stb + tiles + 4 * B].view(torch.int32).cpu().numpy()
    ts.append(t)
    if t > 200 or flags.sum() > 0:
        code = ""
        L_ = _lib.lib()
        if hasattr(L_, "asg_dev_abort_codes"):
            import ctypes
            buf = (ctypes.c_uint * 4)()
            L_.asg_dev_abort_codes(buf)
            code = " abort site %d (count %d, block %d)" % (buf[0], buf[1], buf[2])
        print("call %d: %.0f us, flagged %d %s loss %.4f%s" % (it, t, int(flags.sum()), np.nonzero(flags)[0][:8], float(loss), code))
ts = np.array(ts)
print("median %.1f us, mean %.1f, max %.1f, #>200us: %d" % (np.median(ts), ts.mean(), ts.max(), int((ts > 200).sum())))
