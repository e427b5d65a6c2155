"""CPU tests of the host side of torch_asg_amd (the mirror of the reference's asg.py): the HIP binding is
swapped for a test-only oracle-backed stand-in so defaults, truncation, routing, reductions and the autograd
plumbing can be checked against the golden fixtures without a GPU."""
import numpy as np
import pytest
import torch

import util
from oracle_backend import OracleBackend


@pytest.fixture()
def asg(monkeypatch):
    import torch_asg_amd
    monkeypatch.setattr(torch_asg_amd.asg, "_backend", OracleBackend())
    return torch_asg_amd


def _run(asg, g, dtype, **kw):
    m = asg.ASGLoss(g["transition"].shape[0], reduction=str(g["reduction"]), **kw).to(dtype)
    with torch.no_grad():
        m.transition.copy_(torch.from_numpy(g["transition"]).to(dtype))
    x = torch.from_numpy(g["inputs"]).to(dtype).requires_grad_(True)
    tg = torch.from_numpy(g["targets"])
    if bool(g["pass_lengths"]):
        loss = m(x, tg, torch.from_numpy(g["input_lengths"]), torch.from_numpy(g["target_lengths"]))
    else:
        loss = m(x, tg)
    loss.sum().backward()
    return loss.detach().numpy(), x.grad.numpy(), m.transition.grad.numpy()


@pytest.mark.parametrize("name", util.SMALL)
@pytest.mark.parametrize("kw", [dict(), dict(gpu_no_stream_impl=True)])
def test_host_logic_matches_reference_fixtures(asg, name, kw):
    g = util.load(name)
    loss, gi, gt = _run(asg, g, torch.float64, **kw)
    util.assert_close(loss, g["f64_loss"], 1e-9, name + "/loss")
    util.assert_close(gi, g["f64_grad_inputs"], 1e-9, name + "/gin")
    util.assert_close(gt, g["f64_grad_transition"], 1e-9, name + "/gtr")


def test_eval_and_forward_only_routes(asg):
    g = util.load("cfg1_none")
    args = (torch.from_numpy(g["inputs"]), torch.from_numpy(g["targets"]), torch.from_numpy(g["input_lengths"]),
            torch.from_numpy(g["target_lengths"]))
    for kw, train in ((dict(forward_only=True), True), (dict(), False)):
        m = asg.ASGLoss(7, reduction="none", **kw)
        with torch.no_grad():
            m.transition.copy_(torch.from_numpy(g["transition"]))
        m.train(train)
        out = m(args[0].clone().requires_grad_(True), *args[1:])
        assert not out.requires_grad
        util.assert_close(out.numpy(), g["f32_loss"], 1e-5, "fwd-only")


def test_module_surface_matches_reference():
    import inspect
    import torch_asg_amd
    sig = inspect.signature(torch_asg_amd.ASGLoss.__init__)
    names = list(sig.parameters)
    assert names[:5] == ["self", "num_labels", "reduction", "forward_only", "gpu_no_stream_impl"]
    assert sig.parameters["reduction"].default == "mean"
    assert sig.parameters["forward_only"].default is False and sig.parameters["gpu_no_stream_impl"].default is False
    m = torch_asg_amd.ASGLoss(5)
    assert list(m.state_dict()) == ["transition"] and tuple(m.transition.shape) == (5, 5)
    assert float(m.transition.abs().sum()) == 0.0
    fsig = list(inspect.signature(torch_asg_amd.ASGLoss.forward).parameters)
    assert fsig == ["self", "inputs", "targets", "input_lengths", "target_lengths"]
    for name in ("FAC", "FCC", "ASGGPUFast", "ASGGPUFastForwardOnly"):
        assert hasattr(torch_asg_amd, name)


def test_cpu_tensors_are_rejected_by_the_real_binding():
    import torch_asg_amd
    torch_asg_amd.asg._backend = None
    m = torch_asg_amd.ASGLoss(4)
    with pytest.raises(RuntimeError, match="ROCm device"):
        m(torch.randn(3, 1, 4), torch.zeros(1, 2, dtype=torch.long))


def test_dtype_checks_match_reference_triggers():
    from torch_asg_amd.asg import HipBackend

    class FakeCuda:
        """duck-typed tensor that only answers what _check asks"""
        def __init__(self, dtype, shape, device="cuda:0"):
            self.dtype, self.shape, self.device, self.is_cuda = dtype, shape, device, True

        def dim(self):
            return len(self.shape)

    x = FakeCuda(torch.float32, (3, 1, 4))
    tr = FakeCuda(torch.float32, (4, 4))
    with pytest.raises(RuntimeError, match="Long"):      # utils.cpp:28,46: lengths must be int64
        HipBackend._check(x, tr, None, FakeCuda(torch.int32, (1,)), None)
    with pytest.raises(RuntimeError, match="Float or Double"):
        HipBackend._check(FakeCuda(torch.float16, (3, 1, 4)), tr, None, None, None)
    HipBackend._check(x, tr, FakeCuda(torch.int64, (1, 2)), FakeCuda(torch.int64, (1,)), None)


def test_bench_launcher_and_argument_path_dry_run():
    """bench.py's argument / launcher-environment handling without a GPU: single process, and two ranks under
    torch.distributed.run exactly as the driver launches it (--gpus N must match WORLD_SIZE; rank 0 prints one line)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--steps", "200", "--warmup", "20"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_run"] and d["world"] == 1 and d["steps_per_graph"] == 10 and d["global_batch"] == 64 and not d["uses_dist"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    # the plain way, no launcher: bench.py starts its own two ranks, both reach the collective, ONE line comes back
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", "2"], capture_output=True,
                           text=True, timeout=600, env={k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert plain.returncode == 0, plain.stderr[-2000:]
    lines = [l for l in plain.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, plain.stdout
    d = json.loads(lines[0])
    assert d["world"] == 2 and d["ranks_joined"] == 2 and d["self_launched"] and d["global_batch"] == 128
    # a launcher whose world size contradicts --gpus is still refused
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", "2"], capture_output=True,
                         text=True, timeout=300, env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "20", "--warmup", "4", "--dry-run"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    d = json.loads(lines[0])
    assert d["world"] == 2 and d["global_batch"] == 128 and d["uses_dist"] and d["steps_per_graph"] == 10
    assert d["ranks_joined"] == 2 and not d["self_launched"]


def test_fused_route_policy_and_new_options(asg):
    """The fused training step is taken while every XCD can hold three workgroups for each of its utterances (B <= 80 on 256 CUs); the
    input_is_logits flag is carried by the module."""
    import torch_asg_amd
    from torch_asg_amd.asg import HipBackend
    be = HipBackend()
    be._cus[0] = 256

    class P:
        pass
    dev = torch.device("cuda", 0)
    for Bq, want in ((1, True), (64, True), (80, True), (81, False), (512, False)):
        p = P(); p.B = Bq
        assert be.fused_preferred(p, dev) is want
    m = torch_asg_amd.ASGLoss(5, input_is_logits=True)
    assert m.input_is_logits is True and torch_asg_amd.ASGLoss(5).input_is_logits is False


def test_graphed_refuses_what_it_cannot_record():
    """torch_asg_amd.graphed validates before it touches a device: an ASGLoss module, steps >= 1, a sample that lives on a ROCm device
    (this package has no CPU implementation of the path: a CPU sample is refused like a CPU tensor is by ASGLoss itself)."""
    import pytest
    import torch_asg_amd
    m = torch_asg_amd.ASGLoss(5)
    x, tg = torch.randn(4, 2, 5), torch.zeros(2, 3, dtype=torch.int64)
    with pytest.raises(TypeError, match="ASGLoss"):
        torch_asg_amd.graphed(torch.nn.Linear(2, 2), (x, tg))
    with pytest.raises(ValueError, match="steps"):
        torch_asg_amd.graphed(m, (x, tg), steps=0)
    with pytest.raises(RuntimeError, match="ROCm device"):
        torch_asg_amd.graphed(m, (x, tg))
    assert torch_asg_amd.GraphedStep.replay is torch_asg_amd.GraphedStep.__call__
