#!/usr/bin/env python3
"""Developer probe (GPU): the batched full-lattice forward (asg_batched.hip, forced on with ASG_BATCHED_MIN_B=1) against the
per-utterance chains and the fp64 oracle on a handful of shapes; then step times of both at large batches.
   batched_check.py [check|time|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch_asg_amd, util
from oracle import asg_oracle as orc
dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def run(x, tg, tr, il, tl, red, minb, eval_route=False, mode="single"):
    os.environ["ASG_BATCHED_MIN_B"] = str(minb)
    N = tr.shape[0]
    m = torch_asg_amd.ASGLoss(N, reduction=red, launch_mode=mode).to(dev)
    with torch.no_grad():
        m.transition.copy_(tr)
    xd = x.to(dev).requires_grad_(True)
    if eval_route:
        m.eval()
        with torch.no_grad():
            return dict(loss=m(xd, tg.to(dev), il.to(dev), tl.to(dev)).cpu().numpy())
    loss = m(xd, tg.to(dev), il.to(dev), tl.to(dev))
    fin = torch.isfinite(loss)
    (loss[fin].sum() if red == "none" else loss).backward()
    torch.cuda.synchronize()
    return dict(loss=loss.detach().cpu().numpy(), grad_inputs=xd.grad.cpu().numpy(), grad_transition=m.transition.grad.cpu().numpy())


bad = 0
if what in ("check", "all"):
    rng = np.random.default_rng(5)
    shapes = [(50, 16, 40, 10), (130, 37, 40, 30), (1, 5, 40, 1), (2, 20, 40, 2), (9, 33, 30, 5), (60, 100, 48, 20), (40, 17, 64, 12),
              (33, 48, 8, 4), (70, 19, 33, 9), (45, 64, 44, 11), (25, 40, 16, 6), (90, 21, 52, 13), (17, 16, 24, 5), (400, 64, 40, 30),
              (120, 50, 12, 7), (64, 31, 56, 9), (10, 130, 36, 3), (55, 18, 63, 8)]
    for (T, B, N, L) in shapes:
        for variant in ("plain", "scaled", "neginf", "strided"):
            tr, x, tg, _, _ = util.synth(T, B, N, L, int(rng.integers(0, 1 << 30)))
            il = torch.from_numpy(rng.integers(1, T + 1, B)); tl = torch.from_numpy(rng.integers(1, L + 1, B))
            if rng.random() < 0.5: il[0] = T
            if rng.random() < 0.3: il[:] = T
            if variant == "scaled":
                tr = tr * 30.0 - 10.0
                x = x * 4.0 - 30.0
            if variant == "neginf":
                x[:, :, 1] = float("-inf")
                tg = torch.where(tg == 1, torch.zeros_like(tg), tg)
                tr[2, :] = -300.0
            if variant == "strided":
                x = torch.randn(B, T, N + 3)[:, :, :N].transpose(0, 1)      # batch-major view with a padded label axis
            red = "none"
            o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
            fin = np.isfinite(o["loss"])
            o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red, grad_out=fin.astype(np.float64))
            ra = run(x, tg, tr, il, tl, red, 1)
            rb = run(x, tg, tr, il, tl, red, 1 << 30)
            msg = []
            for k in ("loss", "grad_inputs", "grad_transition"):
                ok, e = util.tol_ok(ra[k], o[k], 1e-4)
                ok2, e2 = util.tol_ok(ra[k], rb[k], 2e-5)
                if not ok or not ok2 or np.isnan(ra[k][np.isfinite(o[k])]).any():
                    msg.append("%s vs oracle %.2e, vs per-utterance %.2e" % (k, e, e2))
            ev = run(x, tg, tr, il, tl, red, 1, eval_route=True)
            ok, e = util.tol_ok(ev["loss"], o["loss"], 1e-4)
            if not ok: msg.append("eval loss %.2e" % e)
            if msg:
                bad += 1
                print("FAIL T=%d B=%d N=%d L=%d %s: %s" % (T, B, N, L, variant, "; ".join(msg)))
    # reduced losses + the other launch modes + determinism
    T, B, N, L = 80, 70, 40, 12
    tr, x, tg, il, tl = util.synth(T, B, N, L, 3, True)
    for red in ("mean", "sum"):
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
        for mode in ("single", "streams", "serial"):
            r = run(x, tg, tr, il, tl, red, 1, mode=mode)
            r2 = run(x, tg, tr, il, tl, red, 1, mode=mode)
            for k in ("loss", "grad_inputs", "grad_transition"):
                ok, e = util.tol_ok(r[k], o[k], 1e-4)
                if not ok or not np.array_equal(r[k], r2[k]):
                    bad += 1
                    print("FAIL reduced %s %s %s: %.2e, repeatable %s" % (red, mode, k, e, np.array_equal(r[k], r2[k])))
    print("check: %d failures" % bad)

if what in ("time", "all"):
    T, N, L = 400, 40, 30
    for B in (512, 1024, 2048, 4096):
        g = torch.Generator().manual_seed(0)
        tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
        tg = torch.randint(0, N, (B, L), generator=g).to(dev)
        il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
        for minb in (1, 1 << 30):
            os.environ["ASG_BATCHED_MIN_B"] = str(minb)
            m = torch_asg_amd.ASGLoss(N).to(dev)
            with torch.no_grad(): m.transition.copy_(tr)
            one = torch.ones((), device=dev)
            def step():
                m.transition.grad = None; x.grad = None
                m(x, tg, il, tl).backward(one)
            def fwd():
                with torch.no_grad():
                    torch_asg_amd.asg.native().loss_forward(x.detach(), tg, m.transition.detach(), il, tl, "mean", torch_asg_amd._lib.FLAG_SINGLE_LAUNCH)
            res = []
            for fn, nrep in ((step, 5), (fwd, 5)):
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for _ in range(3): fn()
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for _ in range(nrep): fn()
                for _ in range(3): gr.replay()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): gr.replay()
                torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 / nrep * 1e6)
            print("B=%5d %-14s step %8.1f us  (%9.0f utt/s)   forward alone %8.1f us" % (B, "batched" if minb == 1 else "per-utterance", res[0], B / res[0] * 1e6, res[1]))
sys.exit(1 if bad else 0)
