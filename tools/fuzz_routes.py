#!/usr/bin/env python3
"""Developer fuzz (GPU): random shapes through the long-target / medium-alphabet routes against the fp64 oracle.
   fuzz_routes.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch_asg_amd, util
from oracle import asg_oracle as orc
dev = "cuda:0"
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t_start = time.time()
for c in range(ncases):
    kind = rng.integers(0, 5)
    big = rng.random() < 0.12          # round 5: beyond 2048 labels (K-sliced streaming step, bfloat16 contraction and its sliced tail, fp64 streaming)
    if kind == 0:      # long targets, small alphabet
        N = int(rng.integers(2, 65)); S = int(rng.integers(65, 1025)); T = int(rng.integers(1, 700))
    elif kind == 1:    # medium alphabet, short targets
        N = int(rng.integers(65, 257)); S = int(rng.integers(1, 65)); T = int(rng.integers(1, 300))
    elif kind == 2:    # both
        N = int(rng.integers(65, 257)); S = int(rng.integers(65, 600)); T = int(rng.integers(1, 400))
    elif kind == 4:    # 257 .. 2048 labels: the matrix resident in a cluster of workgroups (and just beyond it)
        N = int(rng.integers(257, 2100)); S = int(rng.integers(1, 120)); T = int(rng.integers(1, 40))
    else:              # boundaries
        N = int(rng.choice([64, 65, 128, 129, 192, 193, 256, 257])); S = int(rng.choice([64, 65, 128, 129, 256, 257, 512, 513])); T = int(rng.integers(2, 200))
    B = int(rng.integers(1, 5))
    if big:
        # (round 6: batches of more than 64 utterances take the step on the bfloat16 pipe -- fwd_step_bf3, from 1025 labels where the batch rules the
        # resident-slice kernel out)
        N = int(rng.integers(2049, 4600)); S = int(rng.integers(1, 40)); T = int(rng.integers(1, 7)); B = int(rng.choice([1, 2, 3, 33, 40, 65, 70, 97]))
        if B > 64 and rng.random() < 0.5: N = int(rng.integers(1025, 2049))
    junk = torch.full((64 * 1024 * 1024,), float("nan"), device=dev); del junk          # what the allocator hands out next is NaN, not zeros
    dtype = torch.float32 if rng.random() < 0.8 else torch.float64
    tr, x, tg, _, _ = util.synth(T, B, N, S, int(rng.integers(0, 1 << 30)))
    scaled = rng.random() < 0.3
    if scaled:
        tr = tr * float(rng.choice([5.0, 40.0])) - 2.0
    il = rng.integers(1, T + 1, B); tl = rng.integers(1, S + 1, B)
    if rng.random() < 0.5: il[0] = T
    if rng.random() < 0.5: tl[0] = S
    red = ["mean", "sum", "none"][int(rng.integers(0, 3))]
    if (tl > il).any() and red != "none":
        red = "none"
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red)
    m = torch_asg_amd.ASGLoss(N, reduction=red).to(dev).to(dtype)
    with torch.no_grad(): m.transition.copy_(tr.to(dtype))
    xd = x.to(dev, dtype).requires_grad_(True)
    loss = m(xd, tg.to(dev), torch.from_numpy(il).to(dev), torch.from_numpy(tl).to(dev))
    fin = torch.isfinite(loss)
    (loss[fin].sum() if red == "none" else loss).backward() if fin.any() else None
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == torch.float32 else 1e-9      # one gate, scaled transitions included
    res = {"loss": loss.detach().cpu().numpy()}
    if fin.any():
        go = None
        if red == "none":
            go = fin.cpu().numpy().astype(np.float64)
            o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red, grad_out=go)
        res["grad_inputs"] = xd.grad.cpu().numpy(); res["grad_transition"] = m.transition.grad.cpu().numpy()
    msg = []
    for k, v in res.items():
        ok, e = util.tol_ok(v, o[k], tol)
        if not ok or np.isnan(v).any(): msg.append("%s %.2e" % (k, e))
    if msg:
        bad += 1
        if "grad_inputs" in res:
            d = np.abs(res["grad_inputs"].astype(np.float64) - o["grad_inputs"])
            print("   per-utterance max |d grad_inputs|:", ["%.1e" % d[:, bb, :].max() for bb in range(B)],
                  " worst frame of the worst utterance:", int(np.unravel_index(d.argmax(), d.shape)[0]), " max |ref| %.3f" % np.abs(o["grad_inputs"]).max())
        print("FAIL T=%d B=%d N=%d S=%d %s red=%s il=%s tl=%s: %s" % (T, B, N, S, dtype, red, il.tolist(), tl.tolist(), "; ".join(msg)))
print("%d cases, %d failures, %.0f s" % (ncases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
