#!/usr/bin/env python3
"""Developer probe (GPU, library built with -DASG_PROBE_TAIL): per consumer wavefront of utterance 0's alpha workgroup, the
time stamps of its LAST group relative to the end of the recursion."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
from torch_asg_amd import _lib
dev = "cuda:0"
T, B, N, L = 400, 64, 40, 30
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
for _ in range(3):
    loss, saved = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
torch.cuda.synchronize()
ws, gin = saved.tensors; sc, stb, fs = saved.sizes
al = lambda v: (v + 255) // 256 * 256
S = L; off = 0
for sz in (B * T * N * 4, B * T * N * 4, B * T * S * 4, B * T * S * 4, B * T * 2 * 4, N * ((N + 7) // 8 * 8) * 4, N * 4, B * S * 2 * 4, B * S * 2 * 4):      # (asg_api.hip make_layout: ah bh ab bb klog ehat rmax asu asi | dbg)
    off = al(off + sz)
d = ws[sc + off: sc + off + 512].view(torch.int64).cpu().numpy()
end = d[40]
for cw in range(4):
    v = d[cw * 8: cw * 8 + 6]
    print("consumer %d, last group starts at index %d: begins to wait %+d, slot seen %+d, reads+exp done %+d, half-group(0) done %+d, half-group(4) done %+d   (cycles relative to the end of the recursion)"
          % (cw, v[5], v[0] - end, v[1] - end, v[2] - end, v[3] - end, v[4] - end))
print("role ends:", " ".join("wave%d %+d" % (w, d[41 + w] - end) for w in (2, 3, 5, 6)))
