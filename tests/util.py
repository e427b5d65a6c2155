"""Shared helpers for the tests: synthetic inputs, golden loading, tolerance rule."""
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

SMALL = ["cfg1_readme", "cfg1_sum", "cfg1_none", "cfg1_nolengths", "edge_S1", "edge_T1", "edge_T1_S3_trunc",
         "edge_S_gt_T", "edge_infeasible", "edge_il1", "edge_repeats", "edge_tight", "edge_noncontig",
         "peaky", "neginf_label"]
LARGE = ["cfg2", "cfg2_var", "cfg3", "cfg3_var", "cfg5_reduced"]


def synth(T, B, N, L, seed=0, variable=False, dtype=torch.float32):
    """Same draw order as tests/golden/make_golden.py::synth."""
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


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def tol_ok(x, ref, rtol=1e-4):
    """The parity rule of BASELINE.md section 2: max|x-ref| <= rtol * max(1, max|ref|); inf must match exactly."""
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    inf_r, inf_x = np.isinf(ref), np.isinf(x)
    if not np.array_equal(inf_r, inf_x) or not np.array_equal(np.sign(ref[inf_r]), np.sign(x[inf_r])):
        return False, float("inf")
    if np.isnan(x).any():
        return False, float("nan")
    fin = ~inf_r
    if not fin.any():
        return True, 0.0
    scale = max(1.0, float(np.abs(ref[fin]).max()))
    err = float(np.abs(x[fin] - ref[fin]).max())
    return err <= rtol * scale, err / scale


def assert_close(x, ref, rtol=1e-4, what=""):
    ok, e = tol_ok(x, ref, rtol)
    assert ok, "%s: scaled max err %.3e > %.1e" % (what, e, rtol)


def setenv(monkeypatch, name, value):
    """Set (value=None: unset) one of the library's ASG_* developer switches for this test and have the library read them again
    (they are cached after the first call: include/asg_hip.h, asg_reload_env).  The autouse fixture in conftest.py reloads once more
    after the test, when monkeypatch has restored the environment."""
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    from torch_asg_amd import _lib
    _lib.lib().asg_reload_env()
