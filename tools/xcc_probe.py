#!/usr/bin/env python3
"""Developer probe (GPU, library built with -DASG_PROBE_XCC): which XCD the three workgroups of utterances 0..15 ran on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd, bench
from torch_asg_amd import _lib
dev = "cuda:0"
T, B, N, L = 400, 64, 40, 30
tr, x, tg, il, tl = bench.synth(0, dev)
be = torch_asg_amd.asg.native()
for _ in range(2):
    loss, saved = be.loss_forward(x, tg, tr, il, tl, "mean", _lib.FLAG_SINGLE_LAUNCH)
torch.cuda.synchronize()
ws, gin = saved.tensors; sc, stb, fs = saved.sizes
al = lambda v: (v + 255) // 256 * 256
S = L; off = 0
for sz in (B * T * N * 4, B * T * N * 4, B * T * S * 4, B * T * S * 4, N * ((N + 7) // 8 * 8) * 4, N * 4, B * S * 2 * 4, B * S * 2 * 4):
    off = al(off + sz)
d = ws[sc + off: sc + off + 512].view(torch.int64).cpu().numpy()
for b in range(16):
    print("utterance %2d: aligned XCD %d, full-alpha XCD %d, full-beta XCD %d" % (b, d[b * 3], d[b * 3 + 1], d[b * 3 + 2]))
