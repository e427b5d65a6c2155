#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (zh217/torch-asg).

Runs only in the build container: imports the reference's Python package from
/root/reference and its C++ CPU extension from oracle/_ref/ (oracle/build_ref.py).
Each fixture holds inputs and the reference's outputs for
    loss = ASGLoss(N, reduction)(inputs, targets, input_lengths, target_lengths); loss.backward()
in fp32 and fp64 (fp64 = the same inputs up-cast), plus the raw full/aligned scores from
FCC / FAC.  Large cases keep inputs re-generatable from a seed and store only
loss, grad_transition, and a strided sample + checksums of grad_inputs.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.path.insert(0, "/root/reference")

from torch_asg import ASGLoss          # noqa: E402  (the reference)
from torch_asg.asg import FCC, FAC     # noqa: E402


def synth(T, B, N, L, seed=0, variable=False, dtype=torch.float32):
    """SURVEY.md 8(d) synthetic inputs: CPU generator, this exact draw order."""
    g = torch.Generator().manual_seed(seed)
    transition = torch.rand(N, N, generator=g)
    inputs = torch.randn(T, B, N, generator=g)
    targets = torch.randint(0, N, (B, L), generator=g)
    if variable:
        il = torch.randint(T // 2, T + 1, (B,), generator=g)
        tl = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    else:
        il = torch.full((B,), T, dtype=torch.int64)
        tl = torch.full((B,), L, dtype=torch.int64)
    return transition.to(dtype), inputs.to(dtype), targets, il, tl


def run_reference(transition, inputs, targets, il, tl, reduction="mean", dtype=torch.float32,
                  pass_lengths=True):
    N = transition.shape[0]
    m = ASGLoss(N, reduction=reduction)
    m = m.to(dtype)
    with torch.no_grad():
        m.transition.copy_(transition.to(dtype))
    x = inputs.to(dtype).clone().requires_grad_(True)
    if pass_lengths:
        loss = m(x, targets, il, tl)
    else:
        loss = m(x, targets)
    loss.sum().backward()
    # raw scores through the reference's own autograd Functions (test_asg.py:67,219 style)
    T, B, _ = inputs.shape
    S = targets.shape[1]
    tg, tll = targets, tl
    if S > T:
        tg = targets[:, :T]
        tll = torch.clamp(tl, max=T)
    with torch.no_grad():
        full = FCC.apply(transition.to(dtype), inputs.to(dtype), tg, il, tll)
        ali = FAC.apply(transition.to(dtype), inputs.to(dtype), tg, il, tll)
    return dict(loss=loss.detach().numpy(), full_scores=full.numpy(), aligned_scores=ali.numpy(),
                grad_inputs=x.grad.numpy(), grad_transition=m.transition.grad.numpy())


def save_small(name, transition, inputs, targets, il, tl, reduction="mean", pass_lengths=True, note=""):
    out = dict(transition=transition.numpy(), inputs=inputs.numpy(), targets=targets.numpy(),
               input_lengths=il.numpy(), target_lengths=tl.numpy(),
               reduction=np.array(reduction), pass_lengths=np.array(pass_lengths), note=np.array(note))
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        r = run_reference(transition, inputs, targets, il, tl, reduction, dt, pass_lengths)
        for k, v in r.items():
            out["%s_%s" % (tag, k)] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("f64")})


def save_large(name, T, B, N, L, seed, variable, reduction="mean"):
    """Inputs are re-generated from the seed by the tests (synth() is mirrored in tests/util.py)."""
    transition, inputs, targets, il, tl = synth(T, B, N, L, seed, variable)
    out = dict(T=T, B=B, N=N, L=L, seed=seed, variable=variable, reduction=np.array(reduction),
               inputs_checksum=np.float64(inputs.double().sum().item()),
               targets_checksum=np.int64(targets.sum().item()))
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        r = run_reference(transition, inputs, targets, il, tl, reduction, dt)
        gi = r.pop("grad_inputs")
        out["%s_loss" % tag] = r["loss"]
        out["%s_full_scores" % tag] = r["full_scores"]
        out["%s_aligned_scores" % tag] = r["aligned_scores"]
        gt = r["grad_transition"]
        if N <= 64:
            out["%s_grad_transition" % tag] = gt
        else:   # keep the fixture small: strided sample + marginals
            out["%s_grad_transition_sample" % tag] = gt[::8, ::8].copy()
            out["%s_grad_transition_rowsum" % tag] = gt.sum(axis=1)
            out["%s_grad_transition_colsum" % tag] = gt.sum(axis=0)
            out["%s_grad_transition_diag" % tag] = np.diag(gt).copy()
        out["%s_grad_inputs_sample" % tag] = gi[::7, ::3, :].copy()       # strided sample
        out["%s_grad_inputs_sum_t" % tag] = gi.sum(axis=0)                 # [B,N]
        out["%s_grad_inputs_abs_sum" % tag] = np.float64(np.abs(gi.astype(np.float64)).sum())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss32", out["f32_loss"], "loss64", out["f64_loss"])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # cfg 1: README smoke config with variable lengths (BASELINE.json configs[0])
    g = torch.Generator().manual_seed(1)
    T, B, N, L = 6, 2, 7, 5
    tr = torch.rand(N, N, generator=g)
    x = torch.randn(T, B, N, generator=g)
    tg = torch.randint(0, N, (B, L), generator=g)
    il = torch.tensor([6, 4])
    tl = torch.tensor([5, 3])
    save_small("cfg1_readme", tr, x, tg, il, tl, "mean", note="T6 B2 N7 L5 variable lengths")
    save_small("cfg1_sum", tr, x, tg, il, tl, "sum")
    save_small("cfg1_none", tr, x, tg, il, tl, "none")
    save_small("cfg1_nolengths", tr, x, tg, torch.full((B,), T), torch.full((B,), L), "mean",
               pass_lengths=False, note="input_lengths=None,target_lengths=None (asg.py:113-117)")

    # edge cases pinned in SURVEY.md section 4
    def edge(name, T, B, N, L, il, tl, seed, targets=None, reduction="none", note=""):
        g = torch.Generator().manual_seed(seed)
        tr = torch.rand(N, N, generator=g)
        x = torch.randn(T, B, N, generator=g)
        tg = torch.randint(0, N, (B, L), generator=g) if targets is None else targets
        save_small(name, tr, x, tg, torch.tensor(il), torch.tensor(tl), reduction, note=note)

    edge("edge_S1", 5, 2, 4, 1, [5, 3], [1, 1], 2, note="S=1")
    edge("edge_T1", 1, 2, 4, 1, [1, 1], [1, 1], 3, note="T=1")
    edge("edge_T1_S3_trunc", 1, 2, 4, 3, [1, 1], [3, 2], 4, note="T=1,S=3 -> truncation asg.py:119-122")
    edge("edge_S_gt_T", 3, 2, 4, 5, [3, 3], [5, 4], 5, note="S>T truncation")
    edge("edge_infeasible", 6, 3, 5, 4, [6, 2, 5], [3, 4, 4], 6,
         note="target_length > input_length for b=1: loss=+inf, grads NaN-free")
    edge("edge_il1", 5, 2, 4, 2, [1, 5], [1, 2], 7, note="input_length=1")
    edge("edge_repeats", 6, 2, 4, 3, [6, 5], [3, 3], 8, targets=torch.tensor([[1, 1, 1], [2, 2, 0]]),
         note="repeated labels in target")
    edge("edge_tight", 5, 2, 4, 5, [5, 4], [5, 4], 9, note="tl == il (single alignment)")

    # non-contiguous (permuted) inputs like test_asg.py:454 -- stored contiguous; tests re-permute
    g = torch.Generator().manual_seed(10)
    T, B, N, L = 7, 3, 6, 4
    tr = torch.rand(N, N, generator=g)
    xb = torch.randn(B, T, N, generator=g)
    tg = torch.randint(0, N, (B, L), generator=g)
    save_small("edge_noncontig", tr, xb.permute(1, 0, 2), tg, torch.tensor([7, 5, 6]), torch.tensor([4, 2, 3]),
               "none", note="inputs were a permuted [B,T,N] view")

    # large transition magnitudes / peaky emissions (log-softmax of 8*randn)
    g = torch.Generator().manual_seed(11)
    T, B, N, L = 40, 3, 12, 9
    tr = 4.0 * torch.randn(N, N, generator=g)
    x = torch.log_softmax(8.0 * torch.randn(T, B, N, generator=g), dim=-1)
    tg = torch.randint(0, N, (B, L), generator=g)
    save_small("peaky", tr, x, tg, torch.tensor([40, 31, 22]), torch.tensor([9, 7, 9]), "mean",
               note="peaky log-softmax emissions, transitions ~N(0,16)")

    # -inf emissions (masked labels)
    g = torch.Generator().manual_seed(12)
    T, B, N, L = 12, 2, 6, 4
    tr = torch.rand(N, N, generator=g)
    x = torch.randn(T, B, N, generator=g)
    x[:, :, 5] = float("-inf")
    tg = torch.randint(0, 5, (B, L), generator=g)
    save_small("neginf_label", tr, x, tg, torch.tensor([12, 9]), torch.tensor([4, 3]), "mean",
               note="label 5 has -inf emission everywhere")

    # BASELINE configs
    save_large("cfg2", 150, 16, 30, 20, 0, False)
    save_large("cfg2_var", 150, 16, 30, 20, 0, True)
    save_large("cfg3", 400, 64, 40, 30, 0, False)
    save_large("cfg3_var", 400, 64, 40, 30, 0, True)
    save_large("cfg5_reduced", 64, 4, 1024, 16, 0, True)


if __name__ == "__main__":
    main()
