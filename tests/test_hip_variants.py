"""GPU tests (-m gpu) that load DEVELOPER VARIANTS of libasg_hip.so (torch_asg_amd/csrc/build.py::VARIANTS, built by
__graft_entry__.build()) through ASG_HIP_LIB, each in its own process:

  spread   the three workgroups of every utterance of the fused step on three different XCDs: every cross-workgroup
           hand-off (first-half states, aligned verdict / score / edge posteriors, arrival words) crosses L2s.  The
           default placement puts them behind one L2, so the "valid under any placement" claim of DESIGN.md needs this.
  delay    utterance 1's aligned workgroup and utterance 2's full-alpha workgroup start ~0.2 s late, past every bounded
           wait of their partners: the time-out -> flag -> exact redo route, the claim protocol on UttSync::adone, and
           the "sync words are zero again on exit" contract.
  stall    workgroup 1 of every cluster of the resident-slice kernel (256 < N <= 2048) never announces a frame: the
           bounded waits of its peers run out and the scores come back NaN -- the documented failure mode when the grid
           cannot be co-resident (include/asg_hip.h): no hang, no wrong number.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "torch_asg_amd", "csrc", "var_libs")


def _lib(name):
    path = os.path.join(VAR, "libasg_hip_%s.so" % name)
    if not os.path.exists(path):
        pytest.fail("%s is missing: run __graft_entry__.build() (build.py --variants)" % path)
    return path


def _run(args, lib, timeout):
    env = dict(os.environ, ASG_HIP_LIB=lib)
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "variant run failed:\n%s\n%s" % (r.stdout[-4000:], r.stderr[-4000:])
    return r.stdout


def test_parity_suite_with_every_utterance_spread_over_three_xcds():
    out = _run(["-m", "pytest", "tests/test_hip_parity.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                "golden_small_f32 or golden_configs_f32 or random_vs_oracle or every_gradient_element "
                "or determinism or random_shapes_plain_gate or bf16 or fused_step_long or concurrent_calls"], _lib("spread"), 1500)
    assert " passed" in out and "failed" not in out, out[-2000:]


DELAY_SCRIPT = r'''
import sys, time
import numpy as np, torch
sys.path.insert(0, "tests")
import util
from oracle import asg_oracle as orc
import torch_asg_amd
from torch_asg_amd import _lib
assert "delay" in _lib.LIB_PATH
T, B, N, L = 120, 6, 26, 9
tr, x, tg, il, tl = util.synth(T, B, N, L, 31, True)
o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
m = torch_asg_amd.ASGLoss(N, reduction="none").to("cuda:0")
with torch.no_grad():
    m.transition.copy_(tr)
be = torch_asg_amd.asg.native()
res = []
for rep in range(2):                       # the second call reuses the sync region the first one must have left zero
    xd = x.to("cuda:0").requires_grad_(True)
    m.transition.grad = None
    t0 = time.time()
    loss = m(xd, tg.to("cuda:0"), il.to("cuda:0"), tl.to("cuda:0"))
    loss.sum().backward()
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert 0.1 < dt < 20.0, "the delayed workgroups should cost ~0.2 s, not %.3f s" % dt
    for region in be._tickets.values():
        assert int(region.count_nonzero()) == 0, "sync words not left zero"
    res.append((loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), m.transition.grad.cpu().numpy()))
    for got, key in zip(res[-1], ("loss", "grad_inputs", "grad_transition")):
        util.assert_close(got, o[key], 1e-4, "rep %d %s" % (rep, key))
# the flagged utterances went through the exact stand-alone code: same numbers both times
for a, b_ in zip(res[0], res[1]):
    assert np.array_equal(a, b_)
print("DELAY-OK")
'''


def test_late_workgroups_time_out_into_the_exact_redo_without_hanging():
    out = _run(["-c", DELAY_SCRIPT], _lib("delay"), 300)
    assert "DELAY-OK" in out


STALL_SCRIPT = r'''
import sys, time, warnings
import numpy as np, torch
sys.path.insert(0, "tests")
import util
import torch_asg_amd
from torch_asg_amd import _lib
from oracle import asg_oracle as orc
assert "stall" in _lib.LIB_PATH
T, B, N, L = 30, 5, 400, 6
tr, x, tg, il, tl = util.synth(T, B, N, L, 31, True)
o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
m = torch_asg_amd.ASGLoss(N, reduction="none").to("cuda:0")
with torch.no_grad():
    m.transition.copy_(tr)
t0 = time.time()
xd = x.to("cuda:0").requires_grad_(True)
loss = m(xd, tg.to("cuda:0"), il.to("cuda:0"), tl.to("cuda:0"))
loss.sum().backward()
torch.cuda.synchronize()
dt = time.time() - t0
assert dt < 30.0, "the waits are bounded: %.1f s" % dt
# the launch timed out (workgroup 1 of every cluster never publishes a frame in this build) -- and THIS call is right all the same:
# the repair kernel behind it on the stream redid the recursion
assert _lib.lib().asg_cluster_timeouts() >= 1, "the stall variant did not stall"
assert torch.isfinite(loss).all(), loss
util.assert_close(loss.detach().cpu().numpy(), o["loss"], 1e-4, "loss of the call whose launch timed out")
util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs of that call")
util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition of that call")
# the next call says that the resident route is gone (a warning, not an error) and takes the per-frame launches
m.transition.grad = None
xd = x.to("cuda:0").requires_grad_(True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    loss = m(xd, tg.to("cuda:0"), il.to("cuda:0"), tl.to("cuda:0"))
assert any("timed out" in str(v.message) for v in w), [str(v.message) for v in w]
n_before = _lib.lib().asg_cluster_timeouts()
loss.sum().backward()
torch.cuda.synchronize()
assert _lib.lib().asg_cluster_timeouts() == n_before, "the resident route was taken again"
util.assert_close(loss.detach().cpu().numpy(), o["loss"], 1e-4, "loss after the fallback")
util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "grad_inputs after the fallback")
util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "grad_transition after the fallback")
# double precision and the evaluation route repair themselves the same way (a fresh process each: the count is per process)
print("STALL-OK")
'''

STALL_SCRIPT_F64 = r'''
import sys
import numpy as np, torch
sys.path.insert(0, "tests")
import util
import torch_asg_amd
from torch_asg_amd import _lib
from oracle import asg_oracle as orc
T, B, N, L = 24, 20, 300, 5
tr, x, tg, il, tl = util.synth(T, B, N, L, 32, True, torch.float64)
o = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "mean")
m = torch_asg_amd.ASGLoss(N).double().to("cuda:0")
with torch.no_grad():
    m.transition.copy_(tr)
m.eval()
with torch.no_grad():
    ev = m(x.to("cuda:0"), tg.to("cuda:0"), il.to("cuda:0"), tl.to("cuda:0"))
torch.cuda.synchronize()
assert _lib.lib().asg_cluster_timeouts() >= 1
util.assert_close(ev.item(), o["loss"], 1e-9, "evaluation route, fp64, launch timed out")
print("STALL64-OK")
'''


def test_a_cluster_that_cannot_complete_is_repaired_in_the_same_call():
    """never NaN-then-raise-later.  The call whose resident-slice launch runs out of its waits returns the exact
    result (fwd_repair_kernel behind it on the stream), the next call warns and takes the launch-per-frame kernels."""
    out = _run(["-c", STALL_SCRIPT], _lib("stall"), 180)
    assert "STALL-OK" in out, out[-2000:]
    out = _run(["-c", STALL_SCRIPT_F64], _lib("stall"), 180)
    assert "STALL64-OK" in out, out[-2000:]
