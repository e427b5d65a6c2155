"""torch_asg_amd.native_shim: the seven functions of the reference's pybind module `torch_asg_native`
(/root/reference/torch_asg/native/extension.cpp:15-29) on top of libasg_hip.so, so that the reference's own
torch_asg/asg.py runs unmodified.  The GPU tests drive the functions with the argument lists asg.py uses
(asg.py:12-20, 30-32, 42-44, 51-53, 62-64, 76-77, 90-94) and compare with the oracle."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc

DEV = "cuda:0"
# name -> number of positional parameters of the reference's C++ function behind it
# (fully_connected_lattice.h:41-60, force_aligned_lattice.h:42-69, streamlined_fast_gpu.h:17-68)
ARITY = {"fully_connected_forward": 6, "fully_connected_backward": 7, "force_aligned_forward": 9,
         "force_aligned_backward": 11, "fast_asg_gpu_forward_only": 9, "fast_asg_gpu_forward": 9,
         "fast_asg_gpu_backward": 13}


def test_shim_exports_the_seven_pybind_names_with_their_arities():
    import torch_asg_amd.native_shim as shim
    for name, n in ARITY.items():
        f = getattr(shim, name)
        params = [p for p in inspect.signature(f).parameters.values() if p.default is inspect.Parameter.empty]
        assert len(params) == n, (name, len(params), n)


def test_install_makes_the_module_importable_under_the_reference_name():
    import torch_asg_amd.native_shim as shim
    before = sys.modules.get("torch_asg_native")
    try:
        shim.install()
        import torch_asg_native
        assert torch_asg_native is shim
        shim.uninstall()
        assert "torch_asg_native" not in sys.modules
    finally:
        if before is not None:
            sys.modules["torch_asg_native"] = before


@pytest.mark.skipif(not os.path.isdir("/root/reference/torch_asg"), reason="the reference tree is not on this machine")
def test_reference_python_layer_imports_against_the_shim_and_cpu_tensors_are_refused():
    """In the build container (reference present, no GPU): the reference's asg.py, untouched, resolves its
    `import torch_asg_native` to the shim; a CPU call reaches the shim and is refused there (no CPU path)."""
    import torch_asg_amd.native_shim as shim
    saved = {k: sys.modules.get(k) for k in ("torch_asg_native", "torch_asg", "torch_asg.asg")}
    sys.path.insert(0, "/root/reference")
    try:
        for k in ("torch_asg", "torch_asg.asg"):
            sys.modules.pop(k, None)
        shim.install()
        import torch_asg
        assert torch_asg.asg.torch_asg_native is shim
        m = torch_asg.ASGLoss(7)
        with pytest.raises(RuntimeError, match="ROCm device"):
            m(torch.randn(6, 2, 7), torch.randint(0, 7, (2, 5)))
    finally:
        sys.path.remove("/root/reference")
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _case(seed=3, T=37, B=4, N=23, L=9):
    tr, x, tg, il, tl = util.synth(T, B, N, L, seed, True)
    tl = torch.minimum(tl, il)
    return tr, x, tg, il, tl


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
def test_fast_route_call_sequence_matches_the_oracle(dtype, tol):
    """asg.py:71-97: fast_asg_gpu_forward -> six results, the gamma / path_contrib ones handed back to
    fast_asg_gpu_backward together with the upstream gradients of the two scores."""
    import torch_asg_amd.native_shim as nat
    tr, x, tg, il, tl = _case()
    T, B, N = x.shape
    S = tg.shape[1]
    xd, trd = x.to(DEV, dtype), tr.to(DEV, dtype)
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
    res = nat.fast_asg_gpu_forward(xd, tgd, trd, ild, tld, T, B, N, S)
    assert len(res) == 6
    full, ali, g_full, g_ali, pc_full, pc_ali = res
    assert tuple(g_full.shape) == (T, B, N) and tuple(g_ali.shape)[:2] == (T, B) and tuple(full.shape) == (B,)
    w = torch.linspace(0.5, 1.5, B, device=DEV, dtype=dtype)
    gtr, gin = nat.fast_asg_gpu_backward(w, -w, g_full, g_ali, pc_full, pc_ali, tgd, ild, tld, T, B, N, S)
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none",
                     grad_out=w.cpu().double().numpy())
    for name, a, b in (("loss", (full - ali).cpu().numpy(), o["loss"]), ("grad_inputs", gin.cpu().numpy(), o["grad_inputs"]),
                       ("grad_transition", gtr.cpu().numpy(), o["grad_transition"])):
        ok, e = util.tol_ok(a, b, tol)
        assert ok, (name, e)
    only = nat.fast_asg_gpu_forward_only(xd, tgd, trd, ild, tld, T, B, N, S)
    ok, e = util.tol_ok(only.cpu().numpy(), o["loss"], tol)
    assert ok, e


@pytest.mark.gpu
def test_serial_route_call_sequences_match_the_oracle():
    """asg.py:7-55: FCC and FAC through fully_connected_* / force_aligned_*; result = fcc - fac (asg.py:128)."""
    import torch_asg_amd.native_shim as nat
    tr, x, tg, il, tl = _case(seed=5)
    T, B, N = x.shape
    S = tg.shape[1]
    xd, trd = x.to(DEV), tr.to(DEV)
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
    s_full, a_f, b_f, pc_f = nat.fully_connected_forward(xd, trd, ild, T, B, N)
    s_ali, a_a, b_a, pc_a = nat.force_aligned_forward(xd, tgd, trd, ild, tld, T, B, N, S)
    assert tuple(a_f.shape) == (T, B, N) and tuple(a_a.shape) == (T, B, S)
    g = torch.ones(B, device=DEV)
    # two forwards are pending at once (as in the reference's serial route): each backward finds its own state
    gtr_a, gin_a = nat.force_aligned_backward(-g, a_a, b_a, pc_a, tgd, ild, tld, T, B, N, S)
    gtr_f, gin_f = nat.fully_connected_backward(g, a_f, b_f, pc_f, T, B, N)
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    ok, e = util.tol_ok((s_full - s_ali).sum().cpu().numpy(), o["loss"], 1e-4)
    assert ok, e
    ok, e = util.tol_ok((gin_f + gin_a).cpu().numpy(), o["grad_inputs"], 1e-4)
    assert ok, e
    ok, e = util.tol_ok((gtr_f + gtr_a).cpu().numpy(), o["grad_transition"], 1e-4)
    assert ok, e


@pytest.mark.gpu
def test_handles_survive_autograd_saving_and_foreign_tensors_are_refused():
    """The way asg.py uses the results: saved with ctx.save_for_backward inside an autograd.Function, taken back from
    ctx.saved_tensors in backward."""
    import torch_asg_amd.native_shim as nat
    tr, x, tg, il, tl = _case(seed=7)
    T, B, N = x.shape
    S = tg.shape[1]
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)

    class Fast(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inputs, transition):
            full, ali, gf, ga, pf, pa = nat.fast_asg_gpu_forward(inputs, tgd, transition, ild, tld, T, B, N, S)
            ctx.save_for_backward(gf, ga, pf, pa)
            return full, ali

        @staticmethod
        def backward(ctx, g_full, g_ali):
            gf, ga, pf, pa = ctx.saved_tensors
            gtr, gin = nat.fast_asg_gpu_backward(g_full, g_ali, gf, ga, pf, pa, tgd, ild, tld, T, B, N, S)
            return gin, gtr

    xd = x.to(DEV).requires_grad_(True)
    trd = tr.to(DEV).requires_grad_(True)
    full, ali = Fast.apply(xd, trd)
    (full - ali).mean().backward()
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    ok, e = util.tol_ok(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4)
    assert ok, e
    ok, e = util.tol_ok(trd.grad.cpu().numpy(), o["grad_transition"], 1e-4)
    assert ok, e
    stranger = torch.empty(1, device=DEV).expand(T, B, N)
    with pytest.raises(RuntimeError, match="does not come from"):
        nat.fast_asg_gpu_backward(torch.ones(B, device=DEV), -torch.ones(B, device=DEV), stranger, stranger,
                                  stranger, stranger, tgd, ild, tld, T, B, N, S)
    # a uint8 tensor of exactly the right size passes the (synchronisation-free) size test; STRICT also reads the header word
    full, ali, gf, ga, pf, pa = nat.fast_asg_gpu_forward(x.to(DEV), tgd, tr.to(DEV), ild, tld, T, B, N, S)
    lookalike = torch.zeros_like(pf)
    nat.STRICT = True
    try:
        with pytest.raises(RuntimeError, match="does not come from"):
            nat.fast_asg_gpu_backward(torch.ones(B, device=DEV), -torch.ones(B, device=DEV), gf, ga, lookalike, pa, tgd, ild, tld, T, B, N, S)
        gtr, gin = nat.fast_asg_gpu_backward(torch.ones(B, device=DEV) / B, -torch.ones(B, device=DEV) / B, gf, ga, pf, pa, tgd, ild, tld, T, B, N, S)
        ok, e = util.tol_ok(gin.cpu().numpy(), o["grad_inputs"], 1e-4)
        assert ok, e
    finally:
        nat.STRICT = False


@pytest.mark.gpu
def test_saved_tensor_hooks_that_replace_tensors_are_fine():
    """torch.autograd.graph.save_on_cpu moves every saved tensor to the host and hands a NEW device tensor back in backward:
    nothing on this route may be keyed by tensor identity or address -- the state, the transition matrix, the lengths and the
    emissions travel in the data of the tensors asg.py saves."""
    import torch_asg_amd.native_shim as nat
    tr, x, tg, il, tl = _case(seed=11)
    T, B, N = x.shape
    S = tg.shape[1]
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)

    class Serial(torch.autograd.Function):            # FCC of asg.py:37-55
        @staticmethod
        def forward(ctx, transition, inputs):
            scores, alpha, beta, pc = nat.fully_connected_forward(inputs, transition, ild, T, B, N)
            ctx.save_for_backward(alpha, beta, pc)
            return scores

        @staticmethod
        def backward(ctx, g):
            alpha, beta, pc = ctx.saved_tensors
            gtr, gin = nat.fully_connected_backward(g, alpha, beta, pc, T, B, N)
            return gtr, gin

    class Aligned(torch.autograd.Function):           # FAC of asg.py:7-34
        @staticmethod
        def forward(ctx, transition, inputs):
            scores, alpha, beta, pc = nat.force_aligned_forward(inputs, tgd, transition, ild, tld, T, B, N, S)
            ctx.save_for_backward(alpha, beta, pc)
            return scores

        @staticmethod
        def backward(ctx, g):
            alpha, beta, pc = ctx.saved_tensors
            gtr, gin = nat.force_aligned_backward(g, alpha, beta, pc, tgd, ild, tld, T, B, N, S)
            return gtr, gin

    xd = x.to(DEV).requires_grad_(True)
    trd = tr.to(DEV).requires_grad_(True)
    with torch.autograd.graph.save_on_cpu():
        loss = (Serial.apply(trd, xd) - Aligned.apply(trd, xd)).sum()
    loss.backward()
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    for name, a, b in (("loss", loss.detach().cpu().numpy(), o["loss"]), ("grad_inputs", xd.grad.cpu().numpy(), o["grad_inputs"]),
                       ("grad_transition", trd.grad.cpu().numpy(), o["grad_transition"])):
        ok, e = util.tol_ok(a, b, 1e-4)
        assert ok, (name, e)
