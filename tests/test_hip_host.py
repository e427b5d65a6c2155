"""GPU tests (-m gpu) of the host side of the binding: what the autograd Functions keep alive, hipGraph capture of the
fused step, retained graphs, saved-tensor hooks, batch splitting at the 32-bit offset limit, documented exceptions.
Everything is checked against the CPU oracle (tests only) or against another route of the same library."""
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


@pytest.mark.parametrize("mode", ["single", "serial"])
def test_cpu_lengths_on_the_training_route_survive_allocator_churn(mode):
    """The reference takes CPU lengths on its GPU route (streamlined_fast_gpu.cpp:40).  The device copies made for the
    kernels must live until backward: allocate and overwrite a lot of memory between forward and backward."""
    tr, x, tg, il, tl = util.synth(120, 12, 28, 9, 3, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    m = _module(28, tr, launch_mode=mode)
    xd = x.to(DEV).requires_grad_(True)
    loss = m(xd, tg, il[::1].clone(), tl)                        # targets and lengths stay on the CPU
    junk = [torch.full((1 << 16,), 7, dtype=torch.int64, device=DEV) for _ in range(64)]   # reuse freed small blocks
    del junk
    junk = [torch.full((n,), 9, dtype=torch.int64, device=DEV) for n in (12, 12, 64, 128, 12, 12, 256, 12)]
    loss.backward()
    torch.cuda.synchronize()
    del junk
    util.assert_close(loss.item(), o["loss"], 1e-4, "loss")
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition")


def test_noncontiguous_lengths():
    tr, x, tg, il, tl = util.synth(60, 6, 20, 7, 4, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    m = _module(20, tr, reduction="sum")
    il2 = torch.stack([il, il], 1).to(DEV)[:, 0]
    tl2 = torch.stack([tl, tl], 1).to(DEV)[:, 1]
    assert not il2.is_contiguous()
    xd = x.to(DEV).requires_grad_(True)
    m(xd, tg.to(DEV), il2, tl2).backward()
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition")


def test_fused_step_captured_in_a_graph_replays_with_fresh_inputs():
    """hipGraph capture of the fused step (what bench.py times): several steps per graph, replayed with new emissions;
    every replay must equal the eager result on the same values, and eager calls in between must not disturb it."""
    A = _asg()
    T, B, N, L = 200, 24, 40, 14
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True)
    m = _module(N, tr)
    static_x = x.to(DEV).clone().requires_grad_(True)
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                                         # warm-up (creates the sync pool eagerly)
            m.transition.grad = None
            static_x.grad = None
            m(static_x, tgd, ild, tld).backward()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    m.transition.grad = None
    static_x.grad = None
    with torch.cuda.graph(g):
        losses = []
        for _ in range(3):
            loss = m(static_x, tgd, ild, tld)
            loss.backward()
            losses.append(loss)
    for seed in (11, 12, 13):
        _, x2, _, _, _ = util.synth(T, B, N, L, seed, True)
        with torch.no_grad():
            static_x.copy_(x2)
        static_x.grad.zero_()
        m.transition.grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        got = (losses[-1].item(), static_x.grad.clone(), m.transition.grad.clone())
        # an eager call between replays, on the same stream and on another one
        m2 = _module(N, tr)
        xe = x2.to(DEV).requires_grad_(True)
        le = m2(xe, tgd, ild, tld)
        le.backward()
        with torch.cuda.stream(side):
            m3 = _module(N, tr)
            xs = x2.to(DEV).requires_grad_(True)
            m3(xs, tgd, ild, tld).backward()
        torch.cuda.synchronize()
        assert got[0] == le.item()
        assert torch.equal(got[1], 3 * xe.grad), "3 captured steps accumulate 3x the eager gradient"
        assert torch.allclose(got[2], 3 * m2.transition.grad, rtol=1e-6, atol=1e-7)
        assert torch.equal(xs.grad, xe.grad)


def test_first_fused_call_inside_a_capture_is_refused_or_served_from_a_reserved_pool():
    A = _asg()
    be = A.asg.HipBackend()                    # a fresh backend: no pool yet
    old = A.asg._backend
    A.asg._backend = be
    try:
        tr, x, tg, il, tl = util.synth(50, 4, 12, 5, 6, True)
        m = _module(12, tr)
        xd = x.to(DEV).requires_grad_(True)
        tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
        # (the kernels of this alphabet tile have run before: module loading is not part of what is tested)
        old_be = A.asg._backend
        A.asg._backend = old
        m(xd, tgd, ild, tld)
        torch.cuda.synchronize()
        A.asg._backend = old_be
        g = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="warm-up"):
            with torch.cuda.graph(g):
                m(xd, tgd, ild, tld)
        torch.cuda.synchronize()
        be.reserve(DEV)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = m(xd, tgd, ild, tld)
        g.replay()
        torch.cuda.synchronize()
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean", need_grad=False)
        util.assert_close(loss.item(), o["loss"], 1e-4, "loss from a replay")
    finally:
        A.asg._backend = old


@pytest.mark.parametrize("mode", ["single", "serial"])
def test_second_backward_through_a_retained_graph(mode):
    """Fused route: the buffers of the first backward were handed to autograd, the step is recomputed; both passes
    must give the gradients of the split route."""
    tr, x, tg, il, tl = util.synth(90, 10, 25, 8, 7, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    m = _module(25, tr, launch_mode=mode)
    xd = x.to(DEV).requires_grad_(True)
    loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    loss.backward(retain_graph=True)
    g1 = (xd.grad.clone(), m.transition.grad.clone())
    xd.grad = None
    m.transition.grad = None
    (2.0 * loss).backward()
    for got, scale in ((g1, 1.0), ((xd.grad, m.transition.grad), 2.0)):
        util.assert_close(got[0].cpu().numpy(), scale * o["grad_inputs"], 1e-4, "grad_inputs x%g" % scale)
        util.assert_close(got[1].cpu().numpy(), scale * o["grad_transition"], 1e-4, "grad_transition x%g" % scale)


@pytest.mark.parametrize("mode", ["single", "serial"])
def test_saved_tensor_hooks_move_the_saved_buffers(mode):
    """torch.autograd.graph.save_on_cpu: everything saved comes back at other addresses; the problem block is rebuilt."""
    tr, x, tg, il, tl = util.synth(70, 6, 22, 7, 8, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    m = _module(22, tr, launch_mode=mode)
    xd = x.to(DEV).requires_grad_(True)
    with torch.autograd.graph.save_on_cpu():
        loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    junk = torch.full((1 << 20,), 3.0, device=DEV)
    loss.backward()
    del junk
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition")


@pytest.mark.parametrize("kw", [dict(), dict(launch_mode="serial"), dict(gpu_no_stream_impl=True),
                                dict(scale_mode="target_size_sqrt"), dict(reduction="none")])
def test_batches_beyond_the_32bit_offset_limit_are_split(kw):
    """asg_api.hip:check_problem refuses T*B*max(N,S)*w >= 4 GiB on the small-alphabet path; ASGLoss splits such a batch
    along B.  The limit is lowered here so that the split route runs at test size: 3 + 3 + 1 utterances."""
    A = _asg()
    T, B, N, L = 40, 7, 18, 6
    tr, x, tg, il, tl = util.synth(T, B, N, L, 9, True)
    red = kw.get("reduction", "mean")
    m = _module(N, tr, **kw)
    whole = _module(N, tr, **kw)
    m.OFFSET_LIMIT = T * N * 4 * 3 + 1
    assert m._batch_chunk(x, tg) == 3 and whole._batch_chunk(x, tg) == 0
    res = []
    for mod in (m, whole):
        xd = x.to(DEV).requires_grad_(True)
        loss = mod(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
        loss.sum().backward()
        res.append((loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), mod.transition.grad.cpu().numpy()))
    for a, b_, what in zip(res[0], res[1], ("loss", "grad_inputs", "grad_transition")):
        util.assert_close(a, b_, 1e-5, what + " split vs whole")
    if not kw.get("scale_mode"):
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
        util.assert_close(res[0][0], o["loss"], 1e-4, "loss vs oracle")
        util.assert_close(res[0][1], o["grad_inputs"], 1e-4, "grad_inputs vs oracle")


def test_input_is_logits_exception_for_infeasible_utterances():
    """Documented exception of ASGLoss(input_is_logits=True): an utterance with target_length > input_length has
    loss = +inf and, as in the reference, grad_inputs rows that hold only the full-lattice
    posterior -- they sum to g, not to 0 -- so for THAT utterance d loss / d logits differs from d loss / d log-probs
    by softmax * g.  Feasible utterances of the same batch are unaffected."""
    A = _asg()
    T, B, N, L = 12, 3, 9, 8
    tr, x, tg, _, _ = util.synth(T, B, N, L, 10, False, torch.float64)
    il = torch.tensor([12, 5, 12])
    tl = torch.tensor([6, 8, 8])                                   # utterance 1: 8 targets in 5 frames
    m = _module(N, tr.float(), input_is_logits=True, reduction="none")
    ref = _module(N, tr.float(), reduction="none")
    xa = x.float().to(DEV).requires_grad_(True)
    xb = x.float().to(DEV).requires_grad_(True)
    la = m(xa, tg.to(DEV), il.to(DEV), tl.to(DEV))
    lb = ref(torch.log_softmax(xb, 2), tg.to(DEV), il.to(DEV), tl.to(DEV))
    assert torch.isinf(la[1]) and torch.isinf(lb[1]) and torch.isfinite(la[[0, 2]]).all()
    la.sum().backward()
    lb.sum().backward()
    ga, gb = xa.grad.cpu(), xb.grad.cpu()
    assert torch.isfinite(ga).all() and torch.isfinite(gb).all()
    assert torch.allclose(ga[:, [0, 2]], gb[:, [0, 2]], atol=1e-5), "feasible utterances: identical"
    # the infeasible one: rows of the flag route sum to g = 1 over the labels (frames < input_length), and the
    # difference to the composition is exactly softmax * 1
    sm = torch.softmax(x.float(), 2)
    assert torch.allclose(ga[:5, 1].sum(-1), torch.ones(5), atol=1e-5)
    assert torch.allclose(ga[:5, 1] - gb[:5, 1], sm[:5, 1], atol=1e-5)


@pytest.mark.parametrize("shape", [(60, 6, 28, 9), (50, 100, 28, 9), (90, 3, 80, 70), (30, 4, 300, 6)])
@pytest.mark.parametrize("reduction", ["mean", "none"])
def test_cpp_node_python_function_and_python_path_are_the_same_call(shape, reduction):
    """csrc/binding.cpp does in C++ what ASGLossFunction + HipBackend.loss_forward / loss_backward do in Python: same
    buffers, same entry points, same arguments -- so the three host routes (C++ autograd node; Python Function around the
    C++ calls; Python statements + ctypes) must be bit-identical on every kernel route (fused step, stand-alone kernels,
    long targets, large alphabet)."""
    from torch_asg_amd import asg as A
    be = A.native()
    if be.binding is None:
        pytest.skip("torch_asg_amd/_binding.so not built (or ASG_NO_BINDING=1)")
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 11, True)
    m = _module(N, tr, reduction=reduction)
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
    calls = {"apply": 0, "fwd": 0}
    bd = be.binding

    class Spy:
        def __getattr__(self, k):
            return getattr(bd, k)

        def loss_apply(self, *a):
            r = bd.loss_apply(*a)
            calls["apply"] += r is not None
            return r

        def try_loss_forward(self, *a):
            r = bd.try_loss_forward(*a)
            calls["fwd"] += r is not None
            return r
    out, names = [], []
    old_node = A._CPP_NODE
    for route in ("node", "function", "python"):
        be.binding = None if route == "python" else Spy()
        A._CPP_NODE = route == "node"
        try:
            xd = x.to(DEV).requires_grad_(True)
            m.transition.grad = None
            loss = m(xd, tgd, ild, tld)
            names.append(loss.grad_fn.name())
            g = torch.ones_like(loss) * 0.5
            loss.backward(g)
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), xd.grad.clone(), m.transition.grad.clone()))
        finally:
            be.binding = bd
            A._CPP_NODE = old_node
    assert calls == {"apply": 1, "fwd": 1}, "the C++ path declined a plain call: %s" % calls
    assert names[0] == "AsgLossBackward" and names[1] == names[2] == "ASGLossFunctionBackward", names
    deterministic = B <= 80 and N <= 64          # stand-alone / generic routes use atomics-free but order-stable sums too
    for other in out[1:]:
        for a, b in zip(out[0], other):
            if deterministic:
                assert torch.equal(a, b)
            else:
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), reduction)
    assert np.allclose(out[0][0].cpu().numpy(), o["loss"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(60, 6, 28, 9), (50, 100, 28, 9)])
def test_cpp_node_under_the_autograd_engine(shape):
    """What the engine may ask of a grad_fn, asked of AsgLossBackward: a second pass through a retained graph (the fused
    step is recomputed), an error after the buffers were released, only one of the two inputs requiring a gradient, no
    graph under no_grad, an incoming gradient of another dtype / layout, torch.autograd.grad with allow_unused."""
    from torch_asg_amd import asg as A
    be = A.native()
    if be.binding is None or not A._CPP_NODE:
        pytest.skip("C++ autograd node not in use")
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 21, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    w = torch.linspace(0.5, 1.5, B, dtype=torch.float64)
    ow = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none", grad_out=w.numpy())
    m = _module(N, tr, reduction="none")
    tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    loss = m(xd, tgd, ild, tld)
    assert loss.grad_fn.name() == "AsgLossBackward" and loss.requires_grad
    assert [e[0].name() for e in loss.grad_fn.next_functions] == ["torch::autograd::AccumulateGrad"] * 2
    # a float64, non-contiguous incoming gradient
    g64 = torch.stack([w, w], 1).to(DEV)[:, 0]
    loss.backward(g64, retain_graph=True)
    util.assert_close(xd.grad.cpu().numpy(), ow["grad_inputs"], 1e-4, "grad_inputs, weighted")
    util.assert_close(m.transition.grad.cpu().numpy(), ow["grad_transition"], 1e-4, "grad_transition, weighted")
    xd.grad = None
    m.transition.grad = None
    loss.sum().backward()                          # second pass through the retained graph
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs, second pass")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition, second pass")
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        loss.sum().backward()
    # only the transition matrix requires a gradient
    x2 = x.to(DEV)
    m.transition.grad = None
    l2 = m(x2, tgd, ild, tld)
    assert l2.grad_fn is not None
    l2.sum().backward()
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition alone")
    # only the emissions do
    m.transition.requires_grad_(False)
    try:
        x3 = x.to(DEV).requires_grad_(True)
        (gx,) = torch.autograd.grad(m(x3, tgd, ild, tld).sum(), [x3])
        util.assert_close(gx.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs alone")
        assert m(x2, tgd, ild, tld).grad_fn is None
    finally:
        m.transition.requires_grad_(True)
    with torch.no_grad():
        l4 = m(xd, tgd, ild, tld)
    assert l4.grad_fn is None and not l4.requires_grad
    util.assert_close(l4.cpu().numpy(), o["loss"], 1e-4, "loss under no_grad")
    # the node sits in a larger graph
    x5 = x.to(DEV).requires_grad_(True)
    l5 = (m(x5 * 1.0, tgd, ild, tld) * 2.0).sum()
    gx, gt = torch.autograd.grad(l5, [x5, m.transition], allow_unused=True)
    util.assert_close(gx.cpu().numpy(), 2.0 * o["grad_inputs"], 1e-4, "grad_inputs through mul")
    util.assert_close(gt.cpu().numpy(), 2.0 * o["grad_transition"], 1e-4, "grad_transition through mul")


def test_cpp_fast_path_declines_what_python_converts():
    """CPU targets / lengths and strided lengths are not the plain case: the C++ path returns None and the Python path
    converts them (the reference accepts CPU lengths on its GPU route, streamlined_fast_gpu.cpp:40)."""
    from torch_asg_amd import asg as A
    be = A.native()
    if be.binding is None:
        pytest.skip("torch_asg_amd/_binding.so not built (or ASG_NO_BINDING=1)")
    tr, x, tg, il, tl = util.synth(40, 4, 20, 6, 2, True)
    xd, trd = x.to(DEV), tr.to(DEV)
    assert be.binding.try_loss_forward(xd, trd, tg, il.to(DEV), tl.to(DEV), 2, 2) is None           # CPU targets
    il2 = torch.stack([il, il], 1).to(DEV)[:, 0]
    assert be.binding.try_loss_forward(xd, trd, tg.to(DEV), il2, tl.to(DEV), 2, 2) is None          # strided lengths
    assert be.binding.try_loss_forward(xd, trd.double(), tg.to(DEV), None, None, 2, 2) is None      # dtype mismatch
    assert be.binding.try_loss_forward(xd, trd, tg.to(DEV), None, None, 2, 2) is not None
    # the C++ node declines the same, and what ASGLoss.forward settles before the Function: missing lengths, S > T
    assert be.binding.loss_apply(xd, trd, tg, il.to(DEV), tl.to(DEV), 2, 2) is None
    assert be.binding.loss_apply(xd, trd, tg.to(DEV), il2, tl.to(DEV), 2, 2) is None
    assert be.binding.loss_apply(xd, trd, tg.to(DEV), None, None, 2, 2) is None
    long_tg = torch.zeros(4, 41, dtype=torch.int64, device=DEV)
    assert be.binding.loss_apply(xd, trd, long_tg, il.to(DEV), tl.to(DEV), 2, 2) is None
    assert be.binding.loss_apply(xd, trd, tg.to(DEV), il.to(DEV), tl.to(DEV), 2, 2) is not None
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(60, 6, 28, 9), (50, 100, 28, 9), (30, 4, 300, 6)])
def test_sixteen_bit_emissions_are_widened_where_no_kernel_reads_them(dtype, shape):
    """float16 emissions (always) and bfloat16 emissions off the fused training step are widened to the dtype of `transition` inside
    ASGLoss.forward: the result is the float32 result on the same 16-bit-representable values, and inputs.grad has the emissions' dtype."""
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 51, True)
    x16 = x.to(dtype)
    o = orc.asg_loss(x16.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    m = _module(N, tr)
    xd = x16.to(DEV).requires_grad_(True)
    loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    loss.backward()
    assert loss.dtype == torch.float32 and xd.grad.dtype == dtype and m.transition.grad.dtype == torch.float32
    util.assert_close(loss.item(), o["loss"], 1e-4, "loss")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition")
    tol = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7          # (the gradient is rounded to the emissions' dtype on the way out)
    util.assert_close(xd.grad.float().cpu().numpy(), o["grad_inputs"], tol, "grad_inputs")


@pytest.mark.parametrize("shape", [(60, 6, 28, 9), (50, 100, 28, 9), (90, 3, 80, 70), (30, 4, 300, 6), (90, 5, 28, 70)])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_evaluation_route_in_one_call_equals_the_python_route(shape, reduction):
    """module.eval() / forward_only=True: csrc/binding.cpp::Fast.eval_apply runs the beta recursions with `full - aligned` and the
    reduction inside the kernels (asg_loss_forward_only: one launch on the small path); the Python route (ASGGPUFastForwardOnly +
    torch's subtraction and reduction) must give the same numbers, and both the oracle's.  No autograd graph either way."""
    from torch_asg_amd import asg as A
    be = A.native()
    if be.binding is None:
        pytest.skip("torch_asg_amd/_binding.so not built (or ASG_NO_BINDING=1)")
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 61, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), reduction, need_grad=False)
    outs = []
    old = A._CPP_NODE
    for kw, train in ((dict(), False), (dict(forward_only=True), True)):
        m = _module(N, tr, reduction=reduction, **kw)
        m.train(train)
        xd = x.to(DEV).requires_grad_(True)
        for cpp in (True, False):
            A._CPP_NODE = cpp
            try:
                v = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
            finally:
                A._CPP_NODE = old
            assert not v.requires_grad and v.grad_fn is None
            outs.append(v.detach().cpu().numpy())
    for v in outs:
        util.assert_close(v, o["loss"], 1e-4, "evaluation route vs oracle")
    for v in outs[1:]:
        assert np.allclose(v, outs[0], rtol=2e-6, atol=1e-5), "C++ one-call route vs Python route"
    assert np.array_equal(outs[0], outs[2]), "eval() and forward_only=True are the same call"
