"""GPU tests (-m gpu) of `torch_asg_amd.graphed`: the ASGLoss training step recorded into one hipGraph over static buffers
(what bench.py's headline replays).  A replay must BE the eager step on the values in the buffers -- bit for bit on the
fused route -- for fresh emissions, targets and lengths; checked against the CPU oracle too (tests only)."""
import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _asg():
    import torch_asg_amd
    return torch_asg_amd


def _module(N, tr, **kw):
    m = _asg().ASGLoss(N, **kw).to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    return m


def _eager(N, tr, x, tg, il, tl, scale=None, **kw):
    m = _module(N, tr, **kw)
    xd = x.to(DEV).requires_grad_(True)
    loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    g = torch.ones_like(loss) if scale is None else torch.full_like(loss, scale)
    loss.backward(g)
    torch.cuda.synchronize()
    return loss.detach(), xd.grad, m.transition.grad


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_replay_is_the_eager_step_bit_for_bit_with_fresh_values(reduction):
    A = _asg()
    T, B, N, L = 120, 20, 40, 12
    tr, x, tg, il, tl = util.synth(T, B, N, L, 31, True)
    m = _module(N, tr, reduction=reduction)
    step = A.graphed(m, (x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)))
    assert step.inputs_grad is not None and step.loss.shape == (() if reduction != "none" else (B,))
    for seed in (32, 33, 34):
        _, x2, tg2, il2, tl2 = util.synth(T, B, N, L, seed, True)          # fresh emissions, targets AND lengths
        loss = step(x2.to(DEV), tg2.to(DEV), il2.to(DEV), tl2.to(DEV))
        torch.cuda.synchronize()
        le, gx, gt = _eager(N, tr, x2, tg2, il2, tl2, reduction=reduction)
        assert torch.equal(loss, le), "loss"
        assert torch.equal(step.inputs_grad, gx), "inputs_grad"
        assert torch.equal(m.transition.grad, gt), "transition.grad"
        o = orc.asg_loss(x2.double().numpy(), tg2.numpy(), tr.double().numpy(), il2.numpy(), tl2.numpy(), reduction)
        util.assert_close(loss.cpu().numpy(), o["loss"], 1e-4, "loss vs oracle")
        util.assert_close(step.inputs_grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs vs oracle")
        util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition vs oracle")
        # an optimizer that drops the attribute gets it back at the next replay
        m.transition.grad = None
    step()
    assert m.transition.grad is not None


def test_steps_grad_scale_and_writing_into_the_static_buffer():
    A = _asg()
    T, B, N, L = 80, 10, 28, 9
    tr, x, tg, il, tl = util.synth(T, B, N, L, 35, True)
    m = _module(N, tr)
    calls = []
    step = A.graphed(m, (x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)), steps=3, grad_scale=0.25,
                     after_step=lambda: calls.append(1))
    assert len(calls) >= 3                       # called inside the warm-up and the capture, never at replay time
    n = len(calls)
    _, x2, _, _, _ = util.synth(T, B, N, L, 36, True)
    with torch.no_grad():
        step.inputs.copy_(x2)                    # an upstream network writing its output in place
    loss = step()
    torch.cuda.synchronize()
    assert len(calls) == n
    le, gx, gt = _eager(N, tr, x2, tg, il, tl, scale=0.25)
    assert torch.equal(loss, le) and torch.equal(step.inputs_grad, gx) and torch.equal(m.transition.grad, gt)
    with pytest.raises(RuntimeError, match="recorded"):
        step(x2[:, :5].to(DEV))


def test_missing_lengths_and_the_evaluation_route():
    A = _asg()
    T, B, N, L = 60, 8, 20, 7
    tr, x, tg, _, _ = util.synth(T, B, N, L, 37, False)
    m = _module(N, tr)
    step = A.graphed(m, (x.to(DEV), tg.to(DEV)))
    loss = step()
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), None, None, "mean")
    util.assert_close(loss.item(), o["loss"], 1e-4, "loss, default lengths")
    util.assert_close(step.inputs_grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs, default lengths")
    m.eval()
    ev = A.graphed(m, (x.to(DEV), tg.to(DEV)))
    assert ev.inputs_grad is None and not ev.loss.requires_grad
    util.assert_close(ev().item(), o["loss"], 1e-4, "evaluation route")


@pytest.mark.parametrize("shape,env", [((60, 100, 28, 9), None), ((30, 4, 300, 6), None), ((30, 4, 300, 6), "ASG_NO_CLUSTER"),
                                       ((40, 6, 100, 8), None), ((90, 3, 28, 70), None), ((24, 3, 1200, 5), None), ((12, 70, 1100, 4), None)])
def test_every_other_route_replays_too(shape, env, monkeypatch):
    """B > 80 leaves the fused step (recursion kernels + assembly launches), 64 < N <= 256 takes the medium-alphabet kernels, N > 256 the
    resident-slice kernel (or, ASG_NO_CLUSTER=1 / beyond 1024 labels, a launch per frame), S > 64 the long-target kernels: all of them
    record and replay, REPEATEDLY -- every region these routes need zeroed is zeroed by a kernel node (a memset node stops writing zeros at
    the second replay on ROCm 7.2: asg_common.h::zero_async; this test is what found it, through a stand-alone step whose replays kept
    their first loss).  Their sums are order-stable but not bit-pinned against the eager call's allocation pattern: 1e-4 vs the oracle,
    1e-6 vs the eager call."""
    A = _asg()
    if env:
        util.setenv(monkeypatch, env, "1")
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 38, True)
    m = _module(N, tr)
    step = A.graphed(m, (x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)))
    for seed in (39, 40, 41, 42):
        _, x2, tg2, il2, tl2 = util.synth(T, B, N, L, seed, True)
        loss = step(x2.to(DEV), tg2.to(DEV), il2.to(DEV), tl2.to(DEV))
        torch.cuda.synchronize()
        o = orc.asg_loss(x2.double().numpy(), tg2.numpy(), tr.double().numpy(), il2.numpy(), tl2.numpy(), "mean")
        util.assert_close(loss.item(), o["loss"], 1e-4, "loss, replay with seed %d" % seed)
        util.assert_close(step.inputs_grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs, seed %d" % seed)
        util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition, seed %d" % seed)
        le, gx, gt = _eager(N, tr, x2, tg2, il2, tl2)
        assert torch.allclose(loss, le, rtol=1e-6, atol=1e-6)
        assert torch.allclose(step.inputs_grad, gx, rtol=1e-5, atol=1e-6) and torch.allclose(m.transition.grad, gt, rtol=1e-5, atol=1e-5)


def test_make_graphed_callables_takes_the_module():
    """torch.cuda.make_graphed_callables(ASGLoss(...)): PyTorch's own autograd-integrated capture (forward and backward
    as two graphs) -- the C++ autograd node runs under capture on the engine's thread."""
    A = _asg()
    T, B, N, L = 90, 12, 30, 10
    tr, x, tg, il, tl = util.synth(T, B, N, L, 40, True)
    m = _module(N, tr)
    A.reserve(DEV)
    sample = (x.to(DEV).requires_grad_(True), tg.to(DEV), il.to(DEV), tl.to(DEV))
    gm = torch.cuda.make_graphed_callables(m, sample)
    for seed in (41, 42):
        _, x2, tg2, il2, tl2 = util.synth(T, B, N, L, seed, True)
        xd = x2.to(DEV).requires_grad_(True)
        m.transition.grad = None
        loss = gm(xd, tg2.to(DEV), il2.to(DEV), tl2.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        le, gx, gt = _eager(N, tr, x2, tg2, il2, tl2)
        assert torch.equal(loss.detach(), le)
        assert torch.equal(xd.grad, gx) and torch.equal(m.transition.grad, gt)
