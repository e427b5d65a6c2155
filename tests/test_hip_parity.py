"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
(a) golden fixtures from the real reference, (b) the CPU oracle on seeded random inputs,
(c) the reference's own known-answer tests re-typed, (d) size-independent properties.

Tolerance: max|x - ref| <= 1e-4 * max(1, max|ref|) in fp32 -- applied against the
reference's fp64 results (the reference's own fp32 path is noisier than that at T=400);
1e-9 in fp64.  Against the reference's fp32 fixtures the same rule plus the fixture's own fp32-vs-fp64 gap.
"""
import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
MODES = [dict(launch_mode="single"), dict(launch_mode="streams"), dict(launch_mode="serial"),
         dict(gpu_no_stream_impl=True)]


def _asg():
    import torch_asg_amd
    return torch_asg_amd


def run_hip(x, tg, tr, il, tl, reduction, dtype=torch.float32, pass_lengths=True, **kw):
    """x [T,B,N] (any strides, torch CPU), returns dict of numpy arrays."""
    A = _asg()
    N = tr.shape[0]
    m = A.ASGLoss(N, reduction=reduction, **kw).to(DEV).to(dtype)
    with torch.no_grad():
        m.transition.copy_(torch.as_tensor(tr).to(dtype))
    xd = torch.as_tensor(x).to(dtype).to(DEV).requires_grad_(True)
    tgd = torch.as_tensor(tg).to(DEV)
    if pass_lengths:
        loss = m(xd, tgd, torch.as_tensor(il).to(DEV), torch.as_tensor(tl).to(DEV))
    else:
        loss = m(xd, tgd)
    loss.sum().backward()
    torch.cuda.synchronize()
    return dict(loss=loss.detach().cpu().numpy(), grad_inputs=xd.grad.cpu().numpy(),
                grad_transition=m.transition.grad.cpu().numpy())


# ------------------------------------------------------------------ golden, small + edge cases
@pytest.mark.parametrize("name", util.SMALL)
@pytest.mark.parametrize("mode", range(len(MODES)))
def test_golden_small_f32(name, mode):
    g = util.load(name)
    r = run_hip(g["inputs"], g["targets"], g["transition"], g["input_lengths"], g["target_lengths"],
                str(g["reduction"]), torch.float32, bool(g["pass_lengths"]), **MODES[mode])
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], g["f64_" + k], 1e-4, "%s/%s vs ref f64" % (name, k))
        util.assert_close(r[k], g["f32_" + k], 1e-4, "%s/%s vs ref f32" % (name, k))
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()


@pytest.mark.parametrize("name", util.SMALL)
def test_golden_small_f64(name):
    g = util.load(name)
    r = run_hip(g["inputs"], g["targets"], g["transition"], g["input_lengths"], g["target_lengths"],
                str(g["reduction"]), torch.float64, bool(g["pass_lengths"]))
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], g["f64_" + k], 1e-9, "%s/%s" % (name, k))


def test_noncontiguous_inputs():
    # permuted [B,T,N] view as in /root/reference/torch_asg/test/test_asg.py:454
    g = util.load("edge_noncontig")
    xb = torch.from_numpy(g["inputs"]).permute(1, 0, 2).contiguous()     # [B,T,N]
    A = _asg()
    m = A.ASGLoss(g["transition"].shape[0], reduction="none").to(DEV)
    with torch.no_grad():
        m.transition.copy_(torch.from_numpy(g["transition"]))
    xo = xb.to(DEV).requires_grad_(True)
    x = xo.permute(1, 0, 2)
    assert not x.is_contiguous()
    loss = m(x, torch.from_numpy(g["targets"]).to(DEV), torch.from_numpy(g["input_lengths"]).to(DEV),
             torch.from_numpy(g["target_lengths"]).to(DEV))
    loss.sum().backward()
    util.assert_close(loss.detach().cpu().numpy(), g["f64_loss"], 1e-4, "loss")
    util.assert_close(xo.grad.permute(1, 0, 2).cpu().numpy(), g["f64_grad_inputs"], 1e-4, "gin")
    util.assert_close(m.transition.grad.cpu().numpy(), g["f64_grad_transition"], 1e-4, "gtr")


# ------------------------------------------------------------------ golden, BASELINE configs
@pytest.mark.parametrize("name", ["cfg2", "cfg2_var", "cfg3", "cfg3_var"])
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_golden_configs_f32(name, mode):
    g = util.load(name)
    tr, x, tg, il, tl = util.synth(int(g["T"]), int(g["B"]), int(g["N"]), int(g["L"]), int(g["seed"]),
                                   bool(g["variable"]))
    assert abs(float(x.double().sum()) - float(g["inputs_checksum"])) < 1e-6
    r = run_hip(x, tg, tr, il, tl, str(g["reduction"]), torch.float32, **MODES[mode])
    checks = [("loss", r["loss"], "loss"), ("grad_transition", r["grad_transition"], "grad_transition"),
              ("grad_inputs_sample", r["grad_inputs"][::7, ::3, :], "grad_inputs_sample"),
              ("grad_inputs_sum_t", r["grad_inputs"].sum(0), "grad_inputs_sum_t")]
    for what, val, key in checks:
        # hard gate: within 1e-4 (magnitude-relative) of the reference's fp64 answer
        util.assert_close(val, g["f64_" + key], 1e-4, "%s/%s vs ref f64" % (name, what))
        # vs the reference's fp32 answer: 1e-4 plus the reference's own fp32 noise on this quantity
        ok, noise = util.tol_ok(g["f32_" + key], g["f64_" + key], np.inf)
        util.assert_close(val, g["f32_" + key], 1e-4 + noise, "%s/%s vs ref f32" % (name, what))


@pytest.mark.parametrize("name", ["cfg2", "cfg2_var", "cfg3", "cfg3_var"])
def test_configs_every_gradient_element_vs_fp64_oracle(name):
    """The fixtures hold samples of grad_inputs (they must stay small); here EVERY element of both gradients at the
    BASELINE sizes is compared with the fp64 oracle (itself pinned to the same fixtures in tests/test_oracle.py)."""
    g = util.load(name)
    tr, x, tg, il, tl = util.synth(int(g["T"]), int(g["B"]), int(g["N"]), int(g["L"]), int(g["seed"]),
                                   bool(g["variable"]))
    red = str(g["reduction"])
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
    util.assert_close(o["loss"], g["f64_loss"], 1e-9, "oracle vs fixture")
    util.assert_close(o["grad_inputs"][::7, ::3, :], g["f64_grad_inputs_sample"], 1e-9, "oracle vs fixture")
    for kw in (MODES[0], MODES[1]):
        r = run_hip(x, tg, tr, il, tl, red, torch.float32, **kw)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], 1e-4, "%s/%s full tensor vs fp64 oracle" % (name, k))


@pytest.mark.parametrize("name", ["cfg2_var", "cfg3_var"])
def test_golden_configs_f64(name):
    g = util.load(name)
    tr, x, tg, il, tl = util.synth(int(g["T"]), int(g["B"]), int(g["N"]), int(g["L"]), int(g["seed"]),
                                   bool(g["variable"]), torch.float64)
    r = run_hip(x, tg, tr, il, tl, "mean", torch.float64)
    util.assert_close(r["loss"], g["f64_loss"], 1e-10, "loss")
    util.assert_close(r["grad_transition"], g["f64_grad_transition"], 1e-9, "gtr")
    util.assert_close(r["grad_inputs"][::7, ::3, :], g["f64_grad_inputs_sample"], 1e-9, "gin")


# ------------------------------------------------------------------ random shapes vs the oracle
@pytest.mark.parametrize("seed", range(12))
def test_random_vs_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    T = int(rng.integers(1, 70))
    B = int(rng.integers(1, 9))
    N = int(rng.integers(1, 65))
    L = int(rng.integers(1, min(64, T) + 1))
    tr, x, tg, _, _ = util.synth(T, B, N, L, seed)
    il = rng.integers(1, T + 1, B)
    tl = np.minimum(rng.integers(1, L + 1, B), il)
    red = ["mean", "sum", "none"][seed % 3]
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red)
    r = run_hip(x, tg, tr, il, tl, red, torch.float32, **MODES[seed % len(MODES)])
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "seed%d/%s T%d B%d N%d L%d" % (seed, k, T, B, N, L))


@pytest.mark.parametrize("N", [1, 2, 8, 9, 16, 17, 31, 33, 40, 47, 49, 57, 64])
def test_every_alphabet_tile(N):
    T, B, L = 23, 3, min(7, 23)
    tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    r = run_hip(x, tg, tr, il, tl, "sum")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "N%d/%s" % (N, k))


def test_alpha_and_beta_scores_agree():
    """The score read off the end of the alpha pass equals the one read off the end of the beta pass (two independent
    recursions over the same lattice), and both equal the oracle's."""
    A = _asg()
    from torch_asg_amd import _lib
    tr, x, tg, il, tl = util.synth(150, 16, 30, 20, 0, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none",
                     need_grad=False)
    be = A.asg.native()
    full, ali, _ = be.forward(x.to(DEV), tg.to(DEV), tr.to(DEV), il.to(DEV), tl.to(DEV), _lib.FLAG_ALPHA_SCORES)
    full, ali = full.cpu().numpy(), ali.cpu().numpy()
    B = 16
    util.assert_close(full[:B], o["full_scores"], 1e-5, "full beta")
    util.assert_close(full[B:], o["full_scores"], 1e-5, "full alpha")
    util.assert_close(ali[:B], o["aligned_scores"], 1e-5, "aligned beta")
    util.assert_close(ali[B:], o["aligned_scores"], 1e-5, "aligned alpha")


@pytest.mark.parametrize("launch_mode", ["single", "serial", "serial-rowsum"])
@pytest.mark.parametrize("case", ["wide_transitions", "neginf_transitions", "huge_emission_range", "logit_scale"])
def test_exact_fallback_paths(case, launch_mode, monkeypatch):
    """Inputs that push row sums of the exp-domain mat-vec out of fp32 range, so the kernels must take their
    exact log-sum-exp path (forward) / exact softmax path (backward); checked against the fp64 oracle."""
    g = torch.Generator().manual_seed(42)
    T, B, N, L = 37, 3, 19, 6
    tr = torch.rand(N, N, generator=g)
    x = torch.randn(T, B, N, generator=g)
    if case == "wide_transitions":
        tr = 60.0 * torch.randn(N, N, generator=g)                    # span >> 69 nats
    elif case == "neginf_transitions":
        tr = torch.where(torch.rand(N, N, generator=g) < 0.6, torch.full((N, N), float("-inf")), tr)
        tr.fill_diagonal_(0.1)                                        # self loops keep every state alive
        tr[:, 0] = 0.2                                                # and label 0 can go anywhere
    elif case == "huge_emission_range":
        x = 40.0 * torch.randn(T, B, N, generator=g)
    elif case == "logit_scale":
        x = 25.0 + 30.0 * torch.rand(T, B, N, generator=g)            # large positive scores every frame
    tg = torch.randint(0, N, (B, L), generator=g)
    if case == "neginf_transitions":
        tg = torch.zeros(B, L, dtype=torch.long)                      # an alignment that is certainly feasible
    il = torch.tensor([37, 30, 21])
    tl = torch.tensor([6, 4, 5])
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    assert np.isfinite(o["loss"]).all()
    # 'serial' = the stand-alone kernels (the MFMA assembly flags a workgroup and redoes it with the per-frame code: on a NaN in the
    # alpha pass's scale log, or -- 'serial-rowsum', the large-batch kernel -- on a recomputed row sum outside the safe range)
    if launch_mode == "serial-rowsum":
        util.setenv(monkeypatch, "ASG_BWD_ROWSUM", "1")
        launch_mode = "serial"
    r = run_hip(x, tg, tr, il, tl, "none", launch_mode=launch_mode)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "%s/%s/%s" % (case, launch_mode, k))


# ------------------------------------------------------------------ generic path (N > 64 and / or S > 64)
@pytest.mark.parametrize("T,B,N,L", [(20, 3, 100, 7), (33, 5, 257, 12), (150, 2, 12, 100), (100, 3, 130, 80),
                                       (9, 2, 65, 3), (40, 17, 70, 65)])
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
def test_generic_path_vs_oracle(T, B, N, L, dtype, rtol):
    rng = np.random.default_rng(T * 1000 + N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N + L)
    il = rng.integers(max(1, T // 2), T + 1, B)
    tl = np.minimum(rng.integers(1, L + 1, B), il)
    red = ["mean", "sum", "none"][(T + N) % 3]
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red)
    for kw in (MODES[0], MODES[3]):
        r = run_hip(x, tg, tr, il, tl, red, dtype, **kw)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], rtol, "generic T%d B%d N%d L%d %s/%s" % (T, B, N, L, kw, k))


@pytest.mark.parametrize("T,B,N,L", [(30, 3, 257, 6), (25, 20, 300, 5), (20, 40, 600, 4), (14, 5, 1030, 4), (10, 18, 2048, 3), (1, 2, 300, 1)])
def test_f64_alphabets_below_the_streaming_regime(T, B, N, L, monkeypatch):
    """fp64, 256 < N <= 2048 (by default only beyond 1024 since round 5): the per-frame step on v_mfma_f64_16x16x4_f64 (fwd_step_tile_kernel: 16 x 16 output tiles, one or
    two utterance tiles per workgroup, K in contiguous slices per lane group) against the oracle at 1e-9, against the VALU step
    (ASG_NO_TILE_STEP=1) to rounding, run-to-run determinism; variable lengths, an infeasible utterance, the evaluation route."""
    rng = np.random.default_rng(T + N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(1, T // 2), T + 1, B)
    tl = np.minimum(rng.integers(1, L + 1, B), il)
    if B >= 3 and T > 4 and L >= 4:
        il[1], tl[1] = 3, 4                      # infeasible
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    util.setenv(monkeypatch, "ASG_NO_CLUSTER", "1")          # (up to 1024 labels the resident-slice kernel would take the problem)
    outs = []
    for env in ("0", "1"):
        util.setenv(monkeypatch, "ASG_NO_TILE_STEP", env)
        r = run_hip(x, tg, tr, il, tl, "none", torch.float64)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], 1e-9, "fp64 T%d B%d N%d L%d valu=%s/%s" % (T, B, N, L, env, k))
        outs.append(r)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(outs[0][k], outs[1][k], 1e-11, "matrix-core vs VALU step: %s" % k)
    util.setenv(monkeypatch, "ASG_NO_TILE_STEP", "0")
    again = run_hip(x, tg, tr, il, tl, "none", torch.float64)
    assert np.array_equal(again["grad_inputs"], outs[0]["grad_inputs"]) and np.array_equal(again["loss"], outs[0]["loss"], equal_nan=True)
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).double().eval()
    with torch.no_grad():
        m.transition.copy_(tr.double())
        ev = m(x.double().to(DEV), tg.to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV)).cpu().numpy()
    util.assert_close(ev, o["loss"], 1e-9, "fp64 evaluation route")


@pytest.mark.parametrize("no_cluster", ["1", "0"])
def test_f64_matrix_step_exact_path(no_cluster, monkeypatch):
    """Transitions spanning thousands of nats push row sums out of the fp64 exp-domain window (2^+-900) inside
    fwd_step_tile_kernel (ASG_NO_CLUSTER=1) / the fp64 resident-slice kernel: the exact per-node log-sum-exp over the stored
    log-domain state takes over."""
    T, B, N, L = 14, 2, 300, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 9, True)
    tr = tr * 4000.0 - 2000.0
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    util.setenv(monkeypatch, "ASG_NO_CLUSTER", no_cluster)
    r = run_hip(x, tg, tr, il, tl, "none", torch.float64)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-9, "fp64 exact path (ASG_NO_CLUSTER=%s): %s" % (no_cluster, k))
    assert not np.isnan(r["grad_inputs"]).any()


@pytest.mark.parametrize("T,B,N,L", [(7, 3, 2100, 3), (6, 40, 1100, 2), (6, 2, 3300, 2), (5, 34, 6100, 2)])
def test_streaming_step_medium_alphabets(T, B, N, L):
    """fp32 alphabets that take a launch per frame (beyond the resident-slice kernel: N > 2048, or N > 1024 with more than 48
    utterances): fwd_step_mfma's 80-row tiles with one / several row tiles and utterance tiles, against the oracle, variable
    lengths, run-to-run determinism."""
    rng = np.random.default_rng(N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(1, T // 2), T + 1, B)
    tl = np.minimum(rng.integers(1, L + 1, B), il)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "T%d B%d N%d L%d %s" % (T, B, N, L, k))
    r2 = run_hip(x, tg, tr, il, tl, "none")
    assert np.array_equal(r["loss"], r2["loss"]) and np.array_equal(r["grad_inputs"], r2["grad_inputs"])


@pytest.mark.parametrize("T,B,N,L", [(7, 3, 2100, 3), (6, 40, 2500, 2), (5, 34, 3300, 2)])
def test_f64_streaming_step_large_alphabets(T, B, N, L):
    """fp64 beyond 2048 labels: fwd_step_kernel<double> per frame + bwd_post_kernel<double, true> (row sums from the stored
    states, no row-sum contraction) against the oracle at 1e-9; variable lengths, an infeasible utterance, the evaluation
    route, run-to-run determinism.  Reference: double everywhere (/root/reference/torch_asg/native/utils.h:33-39)."""
    rng = np.random.default_rng(T + N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(1, T // 2), T + 1, B)
    tl = np.minimum(rng.integers(1, L + 1, B), il)
    il[1], tl[1] = 1, L                          # infeasible (L >= 2)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    assert np.isinf(o["loss"][1])
    r = run_hip(x, tg, tr, il, tl, "none", torch.float64)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-9, "fp64 T%d B%d N%d L%d %s" % (T, B, N, L, k))
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
    again = run_hip(x, tg, tr, il, tl, "none", torch.float64)
    assert np.array_equal(again["grad_inputs"], r["grad_inputs"]) and np.array_equal(again["loss"], r["loss"])
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).double().eval()
    with torch.no_grad():
        m.transition.copy_(tr.double())
        ev = m(x.double().to(DEV), tg.to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV)).cpu().numpy()
    util.assert_close(ev, o["loss"], 1e-9, "fp64 evaluation route")


def test_streaming_step_at_depth():
    """The per-frame streaming kernels (fwd_step_mfma + bwd_post_kernel + the compacted matrix-core contraction) over 60
    frames -- the kernels of BASELINE.json configs[4] at a depth where the lagged normaliser, the per-frame offsets and the
    compaction of the valid rows have all cycled many times -- against the fp64 oracle at 1e-4 (about a minute of oracle)."""
    T, B, N, L = 60, 34, 2600, 12
    tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "T%d B%d N%d L%d %s" % (T, B, N, L, k))
    r2 = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        assert np.array_equal(r[k], r2[k]), "streaming kernels run to run: " + k


def test_large_alphabet_contraction_on_the_bf16_pipe():
    """Alphabets whose transition gradient is ONE slice of the frame axis (N >= ~2900) take the split-bfloat16 contraction
    (gemm3_pack_kernel + bwd_gemm_bf3_kernel: every operand the exact sum of three bfloat16, six partial products in fp32): against
    the fp64 oracle at 1e-4 over 40 frames with variable lengths (valid rows compacted: K is known on the device only), a label
    count that is no multiple of any tile, run-to-run determinism."""
    T, B, N, L = 40, 20, 3111, 9
    tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "T%d B%d N%d L%d %s" % (T, B, N, L, k))
    ok, err = util.tol_ok(r["grad_transition"], o["grad_transition"], 1e-4)
    assert err < 2e-6, "split-bfloat16 contraction: scaled error %.2e (fp32-equivalent expected)" % err
    r2 = run_hip(x, tg, tr, il, tl, "none")
    assert np.array_equal(r["grad_transition"], r2["grad_transition"])


def test_golden_cfg5_reduced_large_alphabet():
    # BASELINE.json configs[4] at the size the reference can still run: T=64 B=4 N=1024 L=16, variable lengths
    g = util.load("cfg5_reduced")
    tr, x, tg, il, tl = util.synth(64, 4, 1024, 16, 0, True)
    r = run_hip(x, tg, tr, il, tl, "mean")
    util.assert_close(r["loss"], g["f64_loss"], 1e-4, "loss")
    gt = r["grad_transition"]
    util.assert_close(gt[::8, ::8], g["f64_grad_transition_sample"], 1e-4, "gtr sample")
    util.assert_close(gt.sum(1), g["f64_grad_transition_rowsum"], 1e-4, "gtr rowsum")
    util.assert_close(gt.sum(0), g["f64_grad_transition_colsum"], 1e-4, "gtr colsum")
    util.assert_close(np.diag(gt), g["f64_grad_transition_diag"], 1e-4, "gtr diag")
    util.assert_close(r["grad_inputs"][::7, ::3, :], g["f64_grad_inputs_sample"], 1e-4, "gin sample")
    util.assert_close(r["grad_inputs"].sum(0), g["f64_grad_inputs_sum_t"], 1e-4, "gin sum")


@pytest.mark.parametrize("T,B,N,L,il,tl", [(1, 2, 70, 1, [1, 1], [1, 1]), (5, 2, 80, 1, [5, 3], [1, 1]),
                                             (6, 3, 90, 4, [6, 2, 5], [3, 4, 4]), (4, 2, 12, 70, [4, 4], [70, 3])])
def test_generic_edge_cases(T, B, N, L, il, tl):
    # T=1, S=1, infeasible (tl > il -> +inf loss, NaN-free grads), S > T truncation on the wide-target path
    tr, x, tg, _, _ = util.synth(T, B, N, L, 77)
    il, tl = np.array(il), np.array(tl)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "generic edge %s" % k)
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()


def test_unsupported_shapes_fail_loudly():
    A = _asg()
    m = A.ASGLoss(5).to(DEV)
    x = torch.randn(8300, 1, 5, device=DEV)
    tg = torch.zeros(1, 8193, dtype=torch.long, device=DEV)          # targets beyond 8192 positions
    with pytest.raises(RuntimeError, match="unsupported"):
        m(x, tg)


def test_long_targets_over_a_large_alphabet():
    """Targets beyond 1024 positions over more than 2048 labels (refused until round 5: the strip kernel's label scatter was a fixed
    LDS row of 2048 words, its edge scatter an N x N fixed-point image): the row is N words of dynamic LDS, the edges go through the
    hash-table scatter.  Against the fp64 oracle; Viterbi alignment of the same shape."""
    T, B, N, L = 1080, 2, 2100, 1030
    tr, x, tg, _, _ = util.synth(T, B, N, L, 77)
    il = np.array([T, T - 30]); tl = np.array([L, L - 200])
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "long targets over a large alphabet: %s" % k)
    r2 = run_hip(x, tg, tr, il, tl, "none")
    assert np.array_equal(r["grad_transition"], r2["grad_transition"]) and np.array_equal(r["grad_inputs"], r2["grad_inputs"])
    A = _asg()
    m = A.ASGLoss(N).to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    sc, pos, lab = m.viterbi_align(x.to(DEV), tg.to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV))
    osc, opos = orc.viterbi(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl)[:2]
    util.assert_close(sc.cpu().numpy(), osc, 1e-4, "viterbi scores")


def test_generic_forward_only_and_determinism():
    A = _asg()
    tr, x, tg, il, tl = util.synth(30, 3, 90, 70, 4, True)
    tl = torch.minimum(tl, il)
    ref = run_hip(x, tg, tr, il, tl, "none")
    again = run_hip(x, tg, tr, il, tl, "none")
    for k in ref:
        assert np.array_equal(ref[k], again[k]), "generic path run-to-run " + k
    m = A.ASGLoss(90, reduction="none", forward_only=True).to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    out = m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none", need_grad=False)
    util.assert_close(out.cpu().numpy(), o["loss"], 1e-4, "generic forward-only vs oracle")


# ------------------------------------------------------------------ routes
@pytest.mark.parametrize("mode", ["single", "streams", "serial"])
def test_forward_only_and_eval_routes(mode):
    """The beta-only evaluation route (streamlined_fast_gpu.cpp:24-94, routing asg.py:129-131) against the reference's
    fp64 fixtures (cfg2_var / cfg3_var) and against the oracle on a small and on a generic (N > 64, S > 64) shape."""
    A = _asg()

    def route(tr, x, tg, il, tl, N, red, **kw):
        outs = []
        for ctor, train in ((dict(forward_only=True), True), (dict(), False)):
            m = A.ASGLoss(N, reduction=red, launch_mode=mode, **ctor).to(DEV)
            with torch.no_grad():
                m.transition.copy_(tr)
            m.train(train)
            out = m(x.to(DEV).requires_grad_(True), tg.to(DEV), il.to(DEV), tl.to(DEV))
            assert not out.requires_grad            # no backward support on this route (asg.py:66-68)
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])     # forward_only=True and .eval() are the same route
        return outs[0]

    for name in ("cfg2_var", "cfg3_var"):
        g = util.load(name)
        tr, x, tg, il, tl = util.synth(int(g["T"]), int(g["B"]), int(g["N"]), int(g["L"]), int(g["seed"]), True)
        out = route(tr, x, tg, il, tl, int(g["N"]), str(g["reduction"]))
        util.assert_close(out, g["f64_loss"], 1e-4, name + " forward-only vs ref f64")
    for T, B, N, L, seed in ((60, 5, 21, 9, 3), (40, 3, 90, 70, 4)):
        tr, x, tg, il, tl = util.synth(T, B, N, L, seed, True)
        tl = torch.minimum(tl, il)
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none",
                         need_grad=False)
        out = route(tr, x, tg, il, tl, N, "none")
        util.assert_close(out, o["loss"], 1e-4, "forward-only vs oracle T%d N%d L%d" % (T, N, L))


def test_fcc_fac_functions_direct():
    # the reference's tests call FCC / FAC directly with CPU length tensors (test_asg.py:67,219)
    A = _asg()
    tr, x, tg, il, tl = util.synth(20, 4, 11, 6, 5, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    xd = x.to(DEV).requires_grad_(True)
    trd = tr.to(DEV).requires_grad_(True)
    full = A.FCC.apply(trd, xd, tg.to(DEV), il, tl)
    ali = A.FAC.apply(trd, xd, tg.to(DEV), il, tl)
    util.assert_close(full.detach().cpu().numpy(), o["full_scores"], 1e-5, "FCC")
    util.assert_close(ali.detach().cpu().numpy(), o["aligned_scores"], 1e-5, "FAC")
    (full - ali).sum().backward()
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4, "gin")
    util.assert_close(trd.grad.cpu().numpy(), o["grad_transition"], 1e-4, "gtr")
    # each Function alone returns exactly its own part
    xd2 = x.to(DEV).requires_grad_(True)
    A.FCC.apply(tr.to(DEV), xd2, tg.to(DEV), il, tl).sum().backward()
    util.assert_close(xd2.grad.cpu().numpy(), o["grad_inputs_full"], 1e-4, "gin full only")
    xd3 = x.to(DEV).requires_grad_(True)
    (-A.FAC.apply(tr.to(DEV), xd3, tg.to(DEV), il, tl)).sum().backward()
    util.assert_close(xd3.grad.cpu().numpy(), o["grad_inputs_aligned"], 1e-4, "gin aligned only")


def test_fac_alone_over_a_large_alphabet():
    """The force-aligned criterion on its own beyond 2048 labels: grad_transition starts from a memset and receives the edge posteriors
    through the hash-table scatter (no full-lattice gradient before it).  Against the oracle's aligned parts."""
    A = _asg()
    T, B, N, L = 12, 3, 2100, 5
    tr, x, tg, il, tl = util.synth(T, B, N, L, 21, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    xd = x.to(DEV).requires_grad_(True)
    trd = tr.to(DEV).requires_grad_(True)
    ali = A.FAC.apply(trd, xd, tg.to(DEV), il, tl)
    util.assert_close(ali.detach().cpu().numpy(), o["aligned_scores"], 1e-5, "FAC, 2100 labels")
    (-ali).sum().backward()
    util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs_aligned"], 1e-4, "gin aligned only, 2100 labels")
    g = trd.grad.cpu().numpy()
    assert np.isfinite(g).all()
    # the aligned part of grad_transition = total - full part: take the full part from FCC alone
    xd2 = x.to(DEV).requires_grad_(True)
    trd2 = tr.to(DEV).requires_grad_(True)
    A.FCC.apply(trd2, xd2, tg.to(DEV), il, tl).sum().backward()
    util.assert_close(trd2.grad.cpu().numpy() + g, o["grad_transition"], 1e-4, "gtr = full part + aligned part")


def test_determinism_and_mode_equality():
    tr, x, tg, il, tl = util.synth(150, 16, 30, 20, 0, True)
    base = run_hip(x, tg, tr, il, tl, "mean")
    again = run_hip(x, tg, tr, il, tl, "mean")
    for k in base:
        assert np.array_equal(base[k], again[k]), "run-to-run " + k
    # 'streams' and 'serial' run the same stand-alone kernels in different launch arrangements: bit-identical to
    # each other and run to run; 'single' is the fused training step (gradients assembled inside the forward
    # launch, other summation order): equal to rounding
    s1 = run_hip(x, tg, tr, il, tl, "mean", **MODES[1])
    for kw in (MODES[1], MODES[2]):
        r = run_hip(x, tg, tr, il, tl, "mean", **kw)
        for k in s1:
            assert np.array_equal(s1[k], r[k]), "%s differs in mode %s" % (k, kw)
    for k in base:
        util.assert_close(s1[k], base[k], 2e-6, "fused vs stand-alone kernels " + k)
    r = run_hip(x, tg, tr, il, tl, "mean", **MODES[3])   # FCC + FAC summed by autograd: equal to rounding
    for k in base:
        util.assert_close(r[k], base[k], 2e-6, "serial route " + k)


def test_fused_loss_function_equals_reference_style_composition():
    """ASGLoss' fused route (ASGLossFunction) vs the reference-style composition ASGGPUFast + torch ops."""
    A = _asg()
    tr, x, tg, il, tl = util.synth(70, 6, 23, 9, 11, True)
    for red in ("mean", "sum", "none"):
        outs = []
        for fused in (True, False):
            xd = x.to(DEV).requires_grad_(True)
            trd = tr.to(DEV).requires_grad_(True)
            if fused:
                loss = A.ASGLossFunction.apply(xd, trd, tg.to(DEV), il.to(DEV), tl.to(DEV), red, 1)
            else:
                f, a = A.ASGGPUFast.apply(xd, trd, tg.to(DEV), il.to(DEV), tl.to(DEV))
                per = f - a
                loss = per if red == "none" else (per.sum() if red == "sum" else per.mean())
            w = torch.linspace(0.5, 1.5, loss.numel(), device=DEV).reshape(loss.shape)
            (loss * w).sum().backward()
            outs.append((loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), trd.grad.cpu().numpy()))
        for a, b in zip(outs[0], outs[1]):
            util.assert_close(a, b, 2e-6, "fused vs composed (%s)" % red)


def test_transition_grad_accumulates_like_a_parameter():
    A = _asg()
    tr, x, tg, il, tl = util.synth(12, 3, 6, 4, 1, True)
    m = A.ASGLoss(6).to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).backward()
    g1 = m.transition.grad.clone()
    m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).backward()
    assert torch.allclose(m.transition.grad, 2 * g1, rtol=1e-6, atol=1e-7)
    assert "transition" in m.state_dict()


def test_concurrent_calls_on_two_streams_from_two_threads():
    """Re-entrancy of the C ABI (SURVEY.md 8b "Threading"): two ASGLoss forward+backward pairs in flight at the same
    time, on two HIP streams, driven by two host threads, in every launch mode -- bit-identical to the same two
    calls run one after the other.  Everything a call mutates on the device lives in that call's own buffers."""
    import threading
    A = _asg()
    shapes = [(400, 64, 40, 30, 0), (350, 48, 33, 25, 1)]
    probs = [util.synth(T, B, N, L, seed, True) for T, B, N, L, seed in shapes]

    def one(k, mode, stream, out, reps):
        tr, x, tg, il, tl = probs[k]
        N = tr.shape[0]
        with torch.cuda.stream(stream):
            m = A.ASGLoss(N, reduction="mean", launch_mode=mode).to(DEV)
            with torch.no_grad():
                m.transition.copy_(tr)
            xd = x.to(DEV)
            tgd, ild, tld = tg.to(DEV), il.to(DEV), tl.to(DEV)
            res = []
            for _ in range(reps):
                xr = xd.clone().requires_grad_(True)
                m.transition.grad = None
                loss = m(xr, tgd, ild, tld)
                loss.backward()
                res.append((loss.detach().clone(), xr.grad.clone(), m.transition.grad.clone()))
            stream.synchronize()
            out[k] = [tuple(t.cpu().numpy() for t in r) for r in res]

    for mode in ("single", "streams", "serial"):
        seq = {}
        for k in (0, 1):
            one(k, mode, torch.cuda.current_stream(), seq, 1)
        par = {}
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        th = [threading.Thread(target=one, args=(k, mode, streams[k], par, 20)) for k in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in (0, 1):
            for r in par[k]:
                for a, b_ in zip(seq[k][0], r):
                    assert np.array_equal(a, b_), "concurrent call differs from sequential (mode %s)" % mode


# ------------------------------------------------------------------ reference's own tests, re-typed
def test_reference_known_answers_on_gpu():
    import math
    import test_oracle as to
    A = _asg()
    xb, tg, il, tl, loss, gi, gt = to.ASG4()
    for dtype, tol in ((torch.float64, 1e-4), (torch.float32, 1e-4)):
        xo = torch.from_numpy(xb).to(dtype).to(DEV).requires_grad_(True)
        m = A.ASGLoss(6, reduction="none").to(DEV).to(dtype)
        out = m(xo.permute(1, 0, 2), torch.from_numpy(tg).to(DEV), torch.from_numpy(il).to(DEV),
                torch.from_numpy(tl).to(DEV))
        out.sum().backward()
        assert (out.detach().cpu().double().numpy() - loss).__abs__().sum() < 1e-3       # test_asg.py:462
        assert np.abs(xo.grad.cpu().double().numpy() - gi).max() < tol                      # :463
        assert np.abs(m.transition.grad.cpu().double().numpy() - gt).max() < tol            # :464
    # test_asg_2 (:324-351): log 32 ; test_fac_2 (:227-254): -log 32 ; test_fcc_3 (:100-128): S_full = 0
    x = torch.full((3, 1, 4), math.log(0.25), dtype=torch.float64, device=DEV)
    m = A.ASGLoss(4).to(DEV).double()
    assert abs(float(m(x, torch.tensor([[0, 1]], device=DEV), torch.tensor([3]), torch.tensor([2]))) - math.log(32.0)) < 1e-10
    s = A.FAC.apply(torch.zeros(4, 4, dtype=torch.float64, device=DEV), x, torch.tensor([[0, 1]], device=DEV),
                    torch.tensor([3]), torch.tensor([2]))
    assert abs(float(s) + math.log(32.0)) < 1e-10
    g = torch.Generator().manual_seed(0)
    p = torch.rand(3, 300, 40, generator=g)
    xx = torch.log(p / p.sum(-1, keepdim=True)).permute(1, 0, 2).to(DEV)
    s = A.FCC.apply(torch.zeros(40, 40, device=DEV), xx, torch.zeros(3, 50, dtype=torch.long, device=DEV),
                    torch.tensor([300] * 3), torch.tensor([50] * 3))
    assert float(s.abs().sum()) < 1e-4


def test_truncation_and_determinism_asg_3():
    # test_asg.py:354-376: S=4 > T=3 exercises asg.py:119-122; same call twice -> same loss
    import math
    A = _asg()
    x = torch.full((3, 1, 4), math.log(0.25), device=DEV)
    tg = torch.tensor([[0, 1, 1, 1]], device=DEV)
    m = A.ASGLoss(4).to(DEV)
    l1 = m(x, tg, torch.tensor([3], device=DEV), torch.tensor([4], device=DEV))
    l2 = m(x, tg, torch.tensor([3], device=DEV), torch.tensor([4], device=DEV))
    assert float((l1 - l2).abs()) < 1e-10 and math.isfinite(float(l1))


def test_gradcheck_fp64_on_gpu():
    # test_asg.py:131-186,257-288 re-run on the device fp64 kernels
    from torch.autograd import gradcheck
    A = _asg()
    g = torch.Generator().manual_seed(7)
    T, B, N, S = 6, 2, 5, 3
    x = torch.randn(T, B, N, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    tr = torch.rand(N, N, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    tg = torch.randint(0, N, (B, S), generator=g).to(DEV)
    il, tl = torch.tensor([6, 4]), torch.tensor([3, 2])
    assert gradcheck(lambda a, b: A.FCC.apply(b, a, tg, il, tl).sum(), (x, tr))
    assert gradcheck(lambda a, b: A.FAC.apply(b, a, tg, il, tl).sum(), (x, tr))
    assert gradcheck(lambda a, b: torch.stack(A.ASGGPUFast.apply(a, b, tg, il.to(DEV), tl.to(DEV))).sum(0), (x, tr))


# ------------------------------------------------------------------ error behaviour
def test_error_behaviour():
    A = _asg()
    m = A.ASGLoss(4)
    with pytest.raises(RuntimeError):                       # CPU tensors: no fallback
        m(torch.randn(3, 1, 4), torch.zeros(1, 2, dtype=torch.long))
    m = m.to(DEV)
    x = torch.randn(3, 1, 4, device=DEV)
    with pytest.raises(RuntimeError, match="Long"):         # int32 lengths (utils.cpp:28,46)
        m(x, torch.zeros(1, 2, dtype=torch.long, device=DEV), torch.tensor([3], dtype=torch.int32),
          torch.tensor([2], dtype=torch.int32))
    with pytest.raises(RuntimeError):                       # fp16 unsupported
        m.half()(x.half(), torch.zeros(1, 2, dtype=torch.long, device=DEV))


# ------------------------------------------------------------------ properties at full size
def test_cfg3_properties():
    tr, x, tg, il, tl = util.synth(400, 64, 40, 30, 0, True)
    r = run_hip(x, tg, tr, il, tl, "sum")
    gi = r["grad_inputs"]
    ilv = il.numpy()
    for b in range(64):
        assert np.all(gi[ilv[b]:, b, :] == 0), "padded frames must have exactly-zero grads"
        assert np.abs(gi[:ilv[b], b, :].sum(-1)).max() < 2e-5, "posteriors of full and aligned cancel per frame"
    assert abs(r["grad_transition"].astype(np.float64).sum()) <= 1e-5 * np.abs(r["grad_transition"]).sum()
    assert np.isfinite(r["loss"]).all()


def test_cfg4_shard_equals_whole():
    """BASELINE cfg 4 at its stated size on one GPU: T=400 B=512 N=40 L=30 as ONE batch vs 8 `shard_batch` shards of 64
    (SURVEY.md 8e): grad_inputs equal to fp32 rounding (1e-6 absolute on values <= 1), grad_transition and loss equal to
    summation order, and every shard within 1e-4 of an fp64 oracle run of that shard."""
    A = _asg()
    T, B, N, L, W = 400, 512, 40, 30, 8
    tr, x, tg, il, tl = util.synth(T, B, N, L, 2, True)
    whole = run_hip(x, tg, tr, il, tl, "sum")
    acc = np.zeros((N, N), np.float64)
    loss = 0.0
    for r in range(W):
        xs, tgs, ils, tls = A.shard_batch(x, tg, il, tl, r, W)
        assert xs.shape[1] == B // W
        part = run_hip(xs, tgs, tr, ils, tls, "sum")
        acc += part["grad_transition"]
        loss += float(part["loss"])
        lo, hi = r * (B // W), (r + 1) * (B // W)
        # the dispatcher takes the fused training step for 64 utterances and the stand-alone kernels for 512 co-resident
        # chains (same maths, other summation order; the aligned posteriors go through fp32 log-domain states either
        # way): equal to fp32 rounding of values <= 1, not bit for bit
        assert np.abs(part["grad_inputs"] - whole["grad_inputs"][:, lo:hi]).max() < 4e-6
        o = orc.asg_loss(xs.double().numpy(), tgs.numpy(), tr.double().numpy(), ils.numpy(), tls.numpy(), "sum")
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(part[k], o[k], 1e-4, "cfg4 shard %d/%s vs fp64 oracle" % (r, k))
    util.assert_close(acc, whole["grad_transition"], 1e-5, "sharded gtr")
    assert abs(loss - float(whole["loss"])) < 1e-5 * abs(float(whole["loss"]))


def test_cfg5_full_size_properties():
    """BASELINE cfg 5 at FULL size (T=2000 B=32 N=10000 L=60, variable lengths).  The reference cannot run it
    (path_contrib would be 25.6 TB, fully_connected_lattice.cpp:77) and the fp64 oracle would take hours, so parity at
    this size is asserted through size-independent properties of the lattice (value parity of the same kernels is
    checked at the reduced size the reference can run, test_golden_cfg5_reduced_large_alphabet):
      * the score from the alpha pass equals the score from the beta pass (two independent recursions);
      * per valid frame the full-lattice posterior and the aligned posterior both sum to g: their difference sums to 0;
      * frames t >= input_lengths[b] get exactly-zero gradients;
      * every path leaves by as many transitions as the aligned path: sum(grad_transition) == 0 to rounding;
      * the loss is positive (S_full >= S_aligned) and finite."""
    A = _asg()
    from torch_asg_amd import _lib
    T, B, N, L = 2000, 32, 10000, 60
    g = torch.Generator(device=DEV).manual_seed(0)
    tr = torch.rand(N, N, generator=g, device=DEV)
    x = torch.randn(T, B, N, generator=g, device=DEV)
    tg = torch.randint(0, N, (B, L), generator=g, device=DEV)
    il = torch.randint(T // 2, T + 1, (B,), generator=g, device=DEV)
    tl = torch.randint(L // 2, L + 1, (B,), generator=g, device=DEV)
    be = A.asg.native()
    full, ali, st = be.forward(x, tg, tr, il, tl, _lib.FLAG_ALPHA_SCORES)
    fb, fa, ab_, aa = full[:B].double(), full[B:].double(), ali[:B].double(), ali[B:].double()
    assert bool(torch.isfinite(full).all()) and bool(torch.isfinite(ali).all())
    assert float(((fa - fb).abs() / fb.abs()).max()) < 2e-6, "full lattice: alpha score != beta score"
    assert float(((aa - ab_).abs() / ab_.abs()).max()) < 2e-6, "aligned lattice: alpha score != beta score"
    assert bool((fb > ab_).all()), "S_full must exceed S_aligned"
    gf = torch.full((B,), 1.0 / B, device=DEV)
    gtr, gin = be.backward(st, gf, -gf, x, tg, tr, il, tl)
    del st
    assert bool(torch.isfinite(gtr).all()) and bool(torch.isfinite(gin).all())
    valid = torch.arange(T, device=DEV)[:, None] < il[None, :]
    rows = gin.sum(-1)
    assert float(rows[valid].abs().max()) < 1e-4 / B, "posteriors of the two lattices must cancel per frame"
    assert float(gin[~valid].abs().max()) == 0.0, "padded frames must have exactly-zero gradients"
    # |grad_inputs| <= g per element, and the full posterior alone sums to g per frame: spot-check magnitudes
    assert float(gin.abs().max()) <= 1.0 / B * (1 + 1e-5)
    tot, mag = float(gtr.double().sum()), float(gtr.double().abs().sum())
    assert abs(tot) < 1e-5 * mag, (tot, mag)
    # the whole criterion through the module (fused loss route) agrees with the scores above
    m = A.ASGLoss(N, reduction="none").to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    m.eval()
    loss = m(x, tg, il, tl).double()
    assert float(((loss - (fb - ab_)).abs() / (fb - ab_).abs()).max()) < 1e-5


# ------------------------------------------------------------------ best-path (Viterbi) force alignment, SURVEY 8(f)3
def _path_score(x, tr, tg, pos):
    """Score of one alignment (float64 numpy): sum of emissions and of the transitions between consecutive frames."""
    lab = tg[pos]
    s = x[0, lab[0]]
    for t in range(1, len(pos)):
        s += tr[lab[t], lab[t - 1]] + x[t, lab[t]]
    return s


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("T,B,N,L,variable", [(6, 2, 7, 5, True), (23, 5, 9, 7, True), (150, 16, 30, 20, True),
                                              (400, 64, 40, 30, False), (70, 3, 64, 64, True), (1, 2, 4, 1, False),
                                              (150, 3, 30, 100, True), (300, 2, 12, 300, False), (1030, 1, 5, 1000, True),
                                              (1600, 2, 9, 1500, True), (4200, 1, 6, 4096, False), (1100, 3, 40, 1025, True),
                                              (8300, 1, 5, 8192, False), (5200, 2, 7, 5000, True)])
def test_viterbi_vs_oracle(T, B, N, L, variable, dtype):
    A = _asg()
    tr, x, tg, il, tl = util.synth(T, B, N, L, 7, variable, dtype)
    tl = torch.minimum(tl, il)
    sc_o, path_o = orc.viterbi(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy())
    sc, pos, lab = A.viterbi_align(x.to(DEV), tg.to(DEV), tr.to(DEV), il.to(DEV), tl.to(DEV))
    sc, pos, lab = sc.cpu().numpy(), pos.cpu().numpy(), lab.cpu().numpy()
    # adds and compares only, same order as the oracle: bit-exact in both precisions
    assert np.array_equal(sc, sc_o)
    assert np.array_equal(pos, path_o)
    for b in range(B):
        n, o = int(il[b]), int(tl[b])
        p = pos[b, :n]
        assert p[0] == 0 and p[-1] == o - 1 and np.all(np.diff(p) >= 0) and np.all(np.diff(p) <= 1)
        assert np.array_equal(lab[b, :n], tg[b].numpy()[p]) and np.all(lab[b, n:] == -1)
        ref = _path_score(x[:, b].double().numpy(), tr.double().numpy(), tg[b].numpy(), p)
        assert abs(sc[b] - ref) <= (1e-4 if dtype == torch.float32 else 1e-10) * max(1.0, abs(ref))


def test_viterbi_edge_cases_and_module_method():
    A = _asg()
    tr, x, tg, il, tl = util.synth(12, 6, 5, 4, 3, False, torch.float64)
    il = torch.tensor([12, 3, 4, 12, 1, 12])
    tl = torch.tensor([4, 4, 4, 1, 1, 4])                    # utterance 1: more labels than frames -> no alignment
    x = x.clone()
    x[:, 5, :] = -np.inf                                     # utterance 5: nothing emits -> no finite path
    m = A.ASGLoss(5).to(DEV).double()
    with torch.no_grad():
        m.transition.copy_(tr)
    sc, pos, lab = m.viterbi_align(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
    sc_o, path_o = orc.viterbi(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy())
    assert np.array_equal(sc.cpu().numpy(), sc_o) and np.array_equal(pos.cpu().numpy(), path_o)
    assert sc[1] == -np.inf and bool((pos[1] == -1).all()) and sc[5] == -np.inf and bool((pos[5] == -1).all())
    assert pos[2, :4].tolist() == [0, 1, 2, 3] and pos[4, 0] == 0 and not sc.requires_grad
    # best path <= sum over paths (aligned score of the loss), equality when only one alignment exists
    fac = A.FAC.apply(m.transition, x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).detach()
    ok = torch.isfinite(sc)
    assert bool((sc[ok] <= fac[ok] + 1e-9).all()) and abs(float(sc[2] - fac[2])) < 1e-9
    # defaults (lengths None), S > T truncation like ASGLoss.forward, repeated labels, non-contiguous emissions
    xt = torch.randn(4, 3, 6, dtype=torch.float64).transpose(0, 2)      # [T=6,B=3,N=4], label-major strides
    tg2 = torch.tensor([[1, 1, 1, 2, 0, 3, 1, 2]] * 3)
    sc2, pos2, _ = A.viterbi_align(xt.to(DEV), tg2.to(DEV), torch.zeros(4, 4, dtype=torch.float64, device=DEV))
    so, po = orc.viterbi(xt.numpy(), tg2[:, :6].numpy(), np.zeros((4, 4)), None, np.array([6, 6, 6]))
    assert np.array_equal(sc2.cpu().numpy(), so) and np.array_equal(pos2.cpu().numpy(), po)
    with pytest.raises(RuntimeError):
        A.viterbi_align(torch.randn(8300, 1, 4, device=DEV), torch.zeros(1, 8193, dtype=torch.long, device=DEV),
                        torch.zeros(4, 4, device=DEV))        # S > 8192: not supported, fails loudly


def _stress():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_duo", os.path.join(os.path.dirname(__file__), "..", "tools", "stress_duo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [2, 4, 11])
def test_random_shapes_plain_gate_and_determinism(seed):
    """Random (T, B, N, L, lengths) with emissions of ordinary scale (spread <= 5 nats, no common offset; log-probs in
    30% of the cases) and transitions spanning up to 40 nats, against the fp64 oracle under the PLAIN parity rule
    max|x - ref| <= 1e-4 * max(1, max|ref|) per tensor; repeated launches bit-identical.  Exercises the three-wavefront
    chain, its abort-and-redo route and the exact passes."""
    n, worst, _ = _stress().run(seed, 60, regime="plain")
    print("plain-gate battery: %d cases, worst scaled error %.2e" % (n, worst))


@pytest.mark.parametrize("seed", [2, 5])
def test_random_shapes_standalone_route(seed):
    """The same random shapes through launch_mode='serial': the stand-alone recursion kernels and the block-wise
    (matrix-core) gradient assembly, plain parity rule, bit-identical repeats."""
    n, worst, _ = _stress().run(seed, 60, regime="plain", mode="serial")
    assert n == 60 and worst <= 1e-4


@pytest.mark.parametrize("seed", [2, 4, 11])
def test_random_shapes_extended_range_report(seed):
    """REPORTED SEPARATELY from the 1e-4 gate: emissions offset by -40 / +60 and/or spread over 30 nats, where absolute
    scores reach 1e5 and fp32 itself resolves the loss (a difference of two such scores) to ~1e-2.  Gate here:
    determinism, finiteness pattern, and errors within 1e-4 of the magnitude of the quantities the result is a
    difference OF (scores for the loss, sum of lengths for grad_transition), 1e-3 when the spread is 30 nats."""
    n, worst, _ = _stress().run(seed, 60, regime="extended")
    print("extended-range battery: %d cases, worst error (extended rule) %.2e" % (n, worst))


@pytest.mark.parametrize("mode", ["input_size", "input_size_sqrt", "target_size", "target_size_sqrt"])
def test_scale_modes(mode):
    """wav2letter-style per-utterance scaling: loss_b * 1/len or 1/sqrt(len); checked against the oracle's
    unreduced loss and the gradients obtained by feeding the same weights as upstream gradients."""
    A = _asg()
    tr, x, tg, il, tl = util.synth(40, 6, 11, 8, 5, True, torch.float64)
    lens = (il if mode.startswith("input") else tl).double()
    w = 1.0 / (lens.sqrt() if mode.endswith("sqrt") else lens)
    o = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "none", grad_out=w.numpy())
    for red in ("mean", "sum", "none"):
        m = A.ASGLoss(11, reduction=red, scale_mode=mode).to(DEV).double()
        with torch.no_grad():
            m.transition.copy_(tr)
        xd = x.to(DEV).requires_grad_(True)
        loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
        ref = o["loss"] * w.numpy()
        k = 1.0 / 6 if red == "mean" else 1.0
        util.assert_close(loss.detach().cpu().numpy(), ref.mean() if red == "mean" else (ref.sum() if red == "sum" else ref), 1e-9, mode)
        loss.sum().backward()
        util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"] * k, 1e-9, mode + "/gi")
        util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"] * k, 1e-9, mode + "/gt")
    with pytest.raises(ValueError):
        A.ASGLoss(4, scale_mode="bogus")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
def test_input_is_logits(dtype, tol):
    """ASGLoss(input_is_logits=True)(logits) == ASGLoss()(log_softmax(logits)) -- loss and the
    gradients w.r.t. the LOGITS (autograd through torch.log_softmax on the reference-style side), and the loss also
    against the oracle fed log_softmax(logits)."""
    A = _asg()
    tr, x, tg, il, tl = util.synth(90, 7, 33, 12, 21, True, torch.float64)
    x = x * 3.0 + 1.5                                     # unnormalised, with a common offset
    lp = torch.log_softmax(x, dim=2)
    o = orc.asg_loss(lp.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "mean")
    m = A.ASGLoss(33, input_is_logits=True).to(DEV).to(dtype)
    ref = A.ASGLoss(33).to(DEV).to(dtype)
    with torch.no_grad():
        m.transition.copy_(tr.to(dtype))
        ref.transition.copy_(tr.to(dtype))
    xa = x.to(dtype).to(DEV).requires_grad_(True)
    xb = x.to(dtype).to(DEV).requires_grad_(True)
    la = m(xa, tg.to(DEV), il.to(DEV), tl.to(DEV))
    la.backward()
    lb = ref(torch.log_softmax(xb, dim=2), tg.to(DEV), il.to(DEV), tl.to(DEV))
    lb.backward()
    util.assert_close(la.detach().cpu().numpy(), o["loss"], tol, "loss vs oracle(log_softmax)")
    util.assert_close(la.detach().cpu().numpy(), lb.detach().cpu().numpy(), tol, "loss vs composition")
    util.assert_close(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), tol, "d loss / d logits")
    util.assert_close(m.transition.grad.cpu().numpy(), ref.transition.grad.cpu().numpy(), tol, "grad_transition")
    # the oracle's gradient w.r.t. the log-probabilities sums to zero over the labels of every frame: what makes the
    # softmax Jacobian's correction vanish
    assert np.abs(o["grad_inputs"].sum(axis=2)).max() < 1e-12


def test_large_batch_routes_to_standalone_kernels_and_agrees():
    """Above CUs/3 utterances the 'single' launch mode uses the stand-alone kernels (the fused step wants three compute
    units per utterance); both routes must agree with the oracle at a batch on either side of the switch."""
    A = _asg()
    be = A.asg.native()
    for B in (80, 96):           # 3 B <= 256 compute units: fused; above: stand-alone kernels
        tr, x, tg, il, tl = util.synth(40, B, 13, 6, 3, True)
        r = run_hip(x, tg, tr, il, tl, "sum")
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], 1e-4, "B=%d %s" % (B, k))


@pytest.mark.parametrize("rowsum", ["0", "1"])
@pytest.mark.parametrize("T,B,N,L", [(70, 5, 40, 30), (33, 3, 63, 50), (16, 2, 5, 3), (129, 2, 17, 33), (47, 4, 64, 64),
                                     (401, 3, 40, 30), (15, 3, 33, 15), (64, 2, 48, 32), (200, 130, 40, 30)])
def test_standalone_assembly_blocks(T, B, N, L, rowsum, monkeypatch):
    """The stand-alone route's gradient assembly works on 16-frame blocks (full lattice on the matrix cores, aligned
    lattice batched in the same layout): block tails, every label / target-position tile count, variable lengths.
    Both fp32 kernels: row sums from the alpha pass's scale log (ASG_BWD_ROWSUM=0: what small working sets get; B = 130 has the
    one-wavefront chains write the log, the smaller batches the three-wavefront chains) and recomputed (=1: large ones)."""
    util.setenv(monkeypatch, "ASG_BWD_ROWSUM", rowsum)
    tr, x, tg, il, tl = util.synth(T, B, N, L, T + N, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none", launch_mode="serial")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "T%d B%d N%d L%d %s" % (T, B, N, L, k))
    r2 = run_hip(x, tg, tr, il, tl, "none", launch_mode="serial")
    for k in ("loss", "grad_inputs", "grad_transition"):
        assert np.array_equal(r[k], r2[k]), "not deterministic: " + k


@pytest.mark.parametrize("rowsum", ["0", "1"])
@pytest.mark.parametrize("T,B,N,L,mode", [(31, 4, 27, 16, "serial"), (31, 4, 27, 16, "streams"), (50, 100, 40, 12, "single"), (33, 3, 64, 20, "serial")])
def test_standalone_route_emission_offsets(T, B, N, L, mode, rowsum, monkeypatch):
    """Emissions with a common offset of +60 / -40 nats and a spread of 30 (what tools/stress_duo.py draws) through the
    stand-alone kernels, both assembly kernels: frame 0's state is not scaled like the later frames', and an emission factor
    formed for it from the scale log would overflow (found by the stress run, round 4: inf * 0 in the frame without an edge)."""
    util.setenv(monkeypatch, "ASG_BWD_ROWSUM", rowsum)
    for offset, scale in ((60.0, 30.0), (-40.0, 30.0), (60.0, 1.0)):
        tr, x, tg, il, tl = util.synth(T, B, N, L, T + B, True)
        x = x * scale + offset
        tr = (tr - 0.5) * 8.0
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
        r = run_hip(x, tg, tr, il, tl, "none", launch_mode=mode)
        for k in ("loss", "grad_inputs", "grad_transition"):
            assert np.isfinite(r[k][np.isfinite(o[k])]).all(), "%s not finite (offset %g scale %g)" % (k, offset, scale)
            util.assert_close(r[k], o[k], 1e-4, "offset %g scale %g %s/%s" % (offset, scale, mode, k))


def test_very_large_batch_equals_its_chunks():
    """B = 2100 through the stand-alone route (one workgroup per utterance in the assembly, 2100 x 4 recursion chains)
    against the same utterances in chunks of 64 through the fused step: per-utterance losses and input gradients must
    agree, the transition gradient must be the chunks' sum."""
    A = _asg()
    T, B, N, L = 37, 2100, 40, 10
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True)

    def run(sl):
        m = A.ASGLoss(N, reduction="none").to(DEV)
        with torch.no_grad():
            m.transition.copy_(tr)
        xd = x[:, sl].contiguous().to(DEV).requires_grad_(True)
        loss = m(xd, tg[sl].to(DEV), il[sl].to(DEV), tl[sl].to(DEV))
        loss.sum().backward()
        return loss.detach().cpu(), xd.grad.cpu(), m.transition.grad.cpu()

    whole = run(slice(0, B))
    parts = [run(slice(s0, min(s0 + 64, B))) for s0 in range(0, B, 64)]
    ref = (torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts], 1), sum(p_[2] for p_ in parts))
    for name, u, v in zip(("loss", "grad_inputs", "grad_transition"), whole, ref):
        assert torch.isfinite(u).all(), name
        util.assert_close(u.numpy(), v.numpy(), 1e-5, "B=2100 " + name)


def test_fused_step_long_utterances():
    """The fused training step near the top of its supported length (T <= 4000: per-block offset tables of the aligned
    finishers), variable lengths, against the fp64 oracle."""
    tr, x, tg, il, tl = util.synth(3100, 3, 19, 37, 8, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    r = run_hip(x, tg, tr, il, tl, "sum")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "T=3100 %s" % k)


def test_bf16_emissions_fp32_accumulate():
    """SURVEY.md 8(f)2, optional half: bfloat16 emissions in, fp32 arithmetic, bfloat16 gradient out (fused training step).
    Tolerance stated up front: loss and grad_transition against the fp64 oracle fed the SAME bf16-representable values:
    the usual 1e-4 rule; grad_inputs is rounded to bfloat16 on store (8 mantissa bits): 2^-8 = 4e-3 of max(1, max|ref|).
    Also the routes that widen (eval, large batch) must agree with a float32 run on the widened values."""
    A = _asg()
    for (T, B, N, L, seed) in ((120, 9, 33, 14, 5), (400, 64, 40, 30, 0)):
        tr, x, tg, il, tl = util.synth(T, B, N, L, seed, True)
        xb = x.to(torch.bfloat16)
        o = orc.asg_loss(xb.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
        m = A.ASGLoss(N).to(DEV)
        with torch.no_grad():
            m.transition.copy_(tr)
        xd = xb.to(DEV).requires_grad_(True)
        loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
        loss.backward()
        assert loss.dtype == torch.float32 and xd.grad.dtype == torch.bfloat16 and m.transition.grad.dtype == torch.float32
        util.assert_close(loss.detach().cpu().numpy(), o["loss"], 1e-4, "bf16 loss")
        util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "bf16 grad_transition")
        util.assert_close(xd.grad.float().cpu().numpy(), o["grad_inputs"], 2.0 ** -8, "bf16 grad_inputs")
        assert float(xd.grad.float()[int(il.max()):].abs().sum()) == 0.0 if int(il.max()) < T else True
        # widening routes: evaluation, and a batch above the fused limit
        m.eval()
        with torch.no_grad():
            le = m(xb.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
        util.assert_close(le.cpu().numpy(), o["loss"], 1e-4, "bf16 eval loss")
    # flagged utterances (fewer than 4 frames: exact stand-alone redo from the widened copy), fp32 call of the same shape first
    tr, x, tg, _, _ = util.synth(12, 5, 9, 3, 4, False)
    il = torch.tensor([2, 3, 12, 7, 1]); tl = torch.tensor([1, 2, 3, 3, 1])
    run_hip(x, tg, tr, il, tl, "none")
    xb = x.to(torch.bfloat16)
    o = orc.asg_loss(xb.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    m = A.ASGLoss(9, reduction="none").to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    xd = xb.to(DEV).requires_grad_(True)
    lo = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    (lo * torch.linspace(0.5, 1.5, 5, device=DEV)).sum().backward()
    og = orc.asg_loss(xb.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none",
                      grad_out=np.linspace(0.5, 1.5, 5))
    util.assert_close(lo.detach().cpu().numpy(), o["loss"], 1e-4, "bf16 flagged loss")
    util.assert_close(xd.grad.float().cpu().numpy(), og["grad_inputs"], 2.0 ** -8, "bf16 flagged grad_inputs")
    util.assert_close(m.transition.grad.cpu().numpy(), og["grad_transition"], 1e-4, "bf16 flagged grad_transition")
    tr, x, tg, il, tl = util.synth(30, 96, 11, 5, 2, True)
    xb = x.to(torch.bfloat16)
    o = orc.asg_loss(xb.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "sum")
    m = A.ASGLoss(11, reduction="sum").to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    xd = xb.to(DEV).requires_grad_(True)
    m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV)).backward()
    assert xd.grad.dtype == torch.bfloat16
    util.assert_close(xd.grad.float().cpu().numpy(), o["grad_inputs"], 2.0 ** -8, "bf16 grad_inputs (widened route)")
    util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4, "bf16 grad_transition (widened route)")


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(9, 3, 2100, 3), (12, 34, 2500, 4), (7, 32, 4111, 2), (5, 40, 3000, 2), (4, 3, 9000, 2)])
def test_streaming_step_slices_of_k(T, B, N, L):
    """The fp32 streaming step splits K over several workgroups per (row tile, batch tile) when the row tiles alone do not fill the
    device (2 slices at N = 10^4, 8 at N = 2100): the last workgroup to arrive adds the slices in a fixed order.  Against the fp64
    oracle, several batch tiles, a tail chunk (N % 32 != 0), and bit-identical results over repeated runs (the arrival order varies)."""
    tr, x, tg, il, tl = util.synth(T, B, N, L, N, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    first = None
    for rep in range(4):
        r = run_hip(x, tg, tr, il, tl, "none")
        if first is None:
            first = r
            for k in ("loss", "grad_inputs", "grad_transition"):
                util.assert_close(r[k], o[k], 1e-4, "T%d B%d N%d L%d %s" % (T, B, N, L, k))
        else:
            for k in ("loss", "grad_inputs", "grad_transition"):
                assert np.array_equal(first[k], r[k]), "run %d differs in %s" % (rep, k)


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(5, 70, 2200, 2), (4, 64, 5000, 2), (6, 33, 2049, 3), (5, 128, 2300, 2)])
def test_streaming_step_two_batch_tiles(T, B, N, L, monkeypatch):
    """Where the plan prices it lower (ASG_STEP_ONE_TILE=0: whenever B > 32) a workgroup of the fp32 streaming step multiplies its matrix tile
    into TWO batch tiles of 32 utterances (the matrix streamed once for both).  An odd number of batch tiles (the last group's second tile does not exist), a
    second tile of one utterance, four tiles; against the fp64 oracle and against one batch tile per workgroup (ASG_STEP_ONE_TILE=1)."""
    tr, x, tg, il, tl = util.synth(T, B, N, L, N + B, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    util.setenv(monkeypatch, "ASG_STEP_ONE_TILE", 0)
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "two batch tiles T%d B%d N%d %s" % (T, B, N, k))
    util.setenv(monkeypatch, "ASG_STEP_ONE_TILE", 1)
    r1 = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r1[k], o[k], 1e-4, "one batch tile T%d B%d N%d %s" % (T, B, N, k))
        util.assert_close(r1[k], r[k], 1e-5, "one vs two batch tiles T%d B%d N%d %s" % (T, B, N, k))


@pytest.mark.gpu
@pytest.mark.parametrize("mb", [2, 3, 4, 5])
@pytest.mark.parametrize("T,B,N,L", [(6, 3, 2100, 3), (5, 40, 2300, 2), (4, 70, 2600, 2), (5, 2, 4111, 2)])
def test_streaming_step_tile_heights(T, B, N, L, mb, monkeypatch):
    """The fp32 streaming step with 32-, 48-, 64- and 80-row tiles (ASG_STEP_ROW_BLOCKS; the library picks the height that fills the
    device: step_row_blocks): the operand-order copies of the matrix, the K-slice exchange and the epilogue are laid out per
    height.  One and several batch tiles, two tiles per workgroup (B = 70), a last row tile that is mostly padding; against the
    fp64 oracle, and the evaluation route (one direction: a different slice count over the same layout)."""
    tr, x, tg, il, tl = util.synth(T, B, N, L, N + mb, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    util.setenv(monkeypatch, "ASG_STEP_ROW_BLOCKS", mb)
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "%d row blocks T%d B%d N%d %s" % (mb, T, B, N, k))
    r2 = run_hip(x, tg, tr, il, tl, "none")
    assert np.array_equal(r["grad_inputs"], r2["grad_inputs"]) and np.array_equal(r["loss"], r2["loss"])
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).eval()
    with torch.no_grad():
        m.transition.copy_(tr)
        ev = m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).cpu().numpy()
    util.assert_close(ev, o["loss"], 1e-4, "%d row blocks, evaluation route" % mb)


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(6, 16, 2100, 3), (5, 3, 3300, 2), (5, 17, 2100, 2)])
def test_streaming_step_half_tile(T, B, N, L, monkeypatch):
    """Batches of at most 16 utterances: the fp32 streaming step leaves out the second half of its 32-utterance tile (vector loads and
    matrix instructions).  Against the fp64 oracle and bit-identical to the full tile (ASG_STEP_FULL_TILE=1: the first half's
    accumulators see the same products in the same order); B = 17 takes the full tile either way."""
    tr, x, tg, il, tl = util.synth(T, B, N, L, N + B, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "half tile T%d B%d N%d %s" % (T, B, N, k))
    util.setenv(monkeypatch, "ASG_STEP_FULL_TILE", 1)
    r1 = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        assert np.array_equal(r[k], r1[k]), "half vs full tile differ in %s" % k


# ------------------------------------------------------------------ long targets over a small alphabet (letter models)
@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(90, 3, 29, 65), (140, 2, 40, 128), (150, 3, 40, 129), (300, 2, 31, 200),
                                       (280, 2, 64, 256), (330, 2, 8, 300), (520, 2, 40, 512), (600, 1, 5, 513), (1030, 1, 30, 1024)])
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
def test_long_targets_small_alphabet(T, B, N, L, dtype, rtol):
    """64 < S <= 1024 with N <= 64: up to S = 256 one wavefront per aligned chain with 2 / 4 target positions per lane
    (aligned_long_kernel), beyond that a pipeline of wavefronts, one position per thread (aligned_pipe_kernel); label
    scatter through fixed-point LDS rows (bwd_aligned_long_kernel, 2 .. 16 positions per lane).  Small alphabets repeat labels all the time, lengths vary, one utterance is
    infeasible when B >= 3."""
    rng = np.random.default_rng(T + 7 * L)
    tr, x, tg, _, _ = util.synth(T, B, N, L, L + N)
    il = rng.integers(max(L, T // 2), T + 1, B)
    tl = rng.integers(max(1, L - 70), L + 1, B)
    tl[0] = L
    if B >= 3:
        il[2], tl[2] = 20, 40                       # target longer than the input: +inf loss, NaN-free gradients
    red = ["mean", "sum", "none"][(T + N) % 3]
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    r = run_hip(x, tg, tr, il, tl, "none", dtype)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], rtol, "long targets T%d B%d N%d L%d %s" % (T, B, N, L, k))
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
    if B < 3:                                        # (a reduced loss of a batch with an infeasible utterance is +inf)
        o2 = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red)
        r2 = run_hip(x, tg, tr, il, tl, red, dtype, gpu_no_stream_impl=True)     # FCC / FAC separately (asg.py:124-128)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r2[k], o2[k], rtol, "long targets serial T%d L%d %s" % (T, L, k))
    # run-to-run determinism (integer scatter)
    r3 = run_hip(x, tg, tr, il, tl, "none", dtype)
    assert np.array_equal(r["grad_inputs"], r3["grad_inputs"]) and np.array_equal(r["grad_transition"], r3["grad_transition"])


# ------------------------------------------------------------------ 256 < N <= 1024: the matrix resident in a cluster of workgroups
@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(60, 5, 257, 7), (50, 3, 300, 20), (45, 4, 512, 9), (40, 20, 700, 6), (30, 2, 1024, 5),
                                       (12, 70, 1024, 3), (35, 3, 390, 30), (2, 3, 300, 1), (3, 2, 513, 2),
                                       (20, 3, 1025, 4), (16, 20, 1500, 3), (14, 2, 2048, 3), (8, 50, 2000, 2)])
@pytest.mark.parametrize("dtype,rtol,cross", [(torch.float32, 1e-4, 3e-5), (torch.float64, 1e-9, 1e-11)])
def test_resident_slice_alphabets(T, B, N, L, dtype, rtol, cross, monkeypatch):
    """256 < N <= 2048 in fp32 (beyond 1024 labels: up to 16 utterances; (16, 20, 1500, 3) and (8, 50, 2000, 2) take the launch per frame), 256 < N <= 1024 in
    fp64 (v_mfma_f64_16x16x4_f64, workgroups of 512 threads; beyond that the fp64 cases below run the launch per frame both times):
    all frames of the full-lattice recursions in ONE launch (fwd_cluster_kernel: the matrix
    stays in the registers of a cluster of workgroups that exchange the frame's vectors through write-through stores
    and one progress word each).  Stored states, normaliser log and offsets are the per-frame step kernel's, so the
    same gradient pass follows; ASG_NO_CLUSTER=1 (a launch per frame) must agree to rounding, and the route is deterministic.  Variable lengths, an
    infeasible utterance, more chains than one batch of a cluster holds (B = 70 at N = 1024: two rounds)."""
    rng = np.random.default_rng(T + N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(L, T // 2), T + 1, B)
    tl = rng.integers(1, L + 1, B)
    if B >= 3 and T > 4:
        il[1], tl[1] = 3, min(L, 5)                  # infeasible when L >= 4; a 3-frame utterance otherwise
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    outs = []
    for env in ("0", "1"):
        util.setenv(monkeypatch, "ASG_NO_CLUSTER", env)
        r = run_hip(x, tg, tr, il, tl, "none", dtype)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], rtol, "resident slices T%d B%d N%d L%d no_cluster=%s/%s" % (T, B, N, L, env, k))
        assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
        outs.append(r)
    for k in ("loss", "grad_inputs", "grad_transition"):       # (different summation orders: VALU quarters vs MFMA k-steps)
        util.assert_close(outs[0][k], outs[1][k], cross, "cluster vs per-frame launches: %s" % k)
    util.setenv(monkeypatch, "ASG_NO_CLUSTER", "0")
    again = run_hip(x, tg, tr, il, tl, "none", dtype)
    assert np.array_equal(again["grad_inputs"], outs[0]["grad_inputs"]) and np.array_equal(again["loss"], outs[0]["loss"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
def test_resident_slice_cfg3_length(dtype, rtol):
    """fwd_cluster_kernel over cfg 3's frame count and batch (T = 400, B = 64, N = 512): 399 hand-offs per cluster
    (progress words, parity double-buffering) against the fp64 oracle; variable lengths, run-to-run determinism."""
    T, B, N, L = 400, 64, 512, 30
    tr, x, tg, il, tl = util.synth(T, B, N, L, 11, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none", dtype)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], rtol, "resident slices, cfg-3 length: %s" % k)
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
    r2 = run_hip(x, tg, tr, il, tl, "none", dtype)
    assert np.array_equal(r["loss"], r2["loss"]) and np.array_equal(r["grad_inputs"], r2["grad_inputs"])


@pytest.mark.gpu
def test_resident_slice_exact_path_and_eval_route():
    """Transitions spanning hundreds of nats push row sums out of the fp32 exp-domain range inside the cluster kernel:
    the exact log-sum-exp over the other workgroups' stored states takes over; eval route (beta only: one direction)."""
    T, B, N, L = 30, 3, 320, 6
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True)
    tr = tr * 600.0 - 300.0
    tr[3, :] = float("-inf")
    tr[3, 3] = 0.0
    tr[:, 7] = -250.0
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "resident slices exact path %s" % k)
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).eval()
    with torch.no_grad():
        m.transition.copy_(tr)
        ev = m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).cpu().numpy()
    util.assert_close(ev, o["loss"], 1e-4, "resident slices eval route")


# ------------------------------------------------------------------ medium alphabets (64 < N <= 256): one launch per pass
@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L", [(60, 3, 65, 9), (120, 2, 128, 20), (45, 4, 129, 7), (80, 2, 192, 30), (70, 3, 200, 70),
                                       (50, 2, 256, 11), (260, 2, 130, 200), (300, 2, 66, 290)])
def test_medium_alphabets(T, B, N, L):
    """fp32, 64 < N <= 256: the full-lattice recursions run as ONE launch (fwd_mid_kernel: a workgroup of ceil(N / 64)
    wavefronts per chain, transition row in registers), the gradient through the row-sum and outer-product contractions
    with the frame axis split over workgroups; ASG_NO_MID=1 (per-frame launches) must agree to rounding."""
    rng = np.random.default_rng(T + N)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(L, T // 2), T + 1, B)
    tl = rng.integers(1, L + 1, B)
    if B >= 3:
        il[1], tl[1] = 3, min(L, 5)                  # infeasible utterance
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    for kw in (MODES[0], MODES[3]):
        r = run_hip(x, tg, tr, il, tl, "none", **kw)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(r[k], o[k], 1e-4, "medium alphabet T%d B%d N%d L%d %s/%s" % (T, B, N, L, kw, k))
        assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
    r2 = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):      # serial (FCC / FAC entry points) and single share the kernels: equal to rounding
        util.assert_close(r[k], r2[k], 2e-6, "medium alphabet serial vs single: %s" % k)
    a = run_hip(x, tg, tr, il, tl, "none")
    b_ = run_hip(x, tg, tr, il, tl, "none")
    assert np.array_equal(a["grad_inputs"], b_["grad_inputs"]) and np.array_equal(a["grad_transition"], b_["grad_transition"])


@pytest.mark.gpu
def test_medium_alphabet_exact_path_and_eval_route():
    """Transitions spanning hundreds of nats (and -inf entries) push row sums out of the fp32 exp-domain range: the
    per-node exact log-sum-exp takes over; the evaluation route (beta only) uses the same kernel."""
    T, B, N, L = 40, 2, 100, 6
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True)
    tr = tr * 600.0 - 300.0
    tr[3, :] = float("-inf")
    tr[3, 3] = 0.0
    tr[:, 7] = -250.0
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    r = run_hip(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "medium alphabet exact path %s" % k)
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).eval()
    with torch.no_grad():
        m.transition.copy_(tr)
        ev = m(x.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
    util.assert_close(ev.cpu().numpy(), o["loss"], 1e-4, "medium alphabet eval")


# ------------------------------------------------------------------ very long targets (1024 < S <= 8192)
@pytest.mark.gpu
@pytest.mark.parametrize("T,B,N,L,dtype,rtol", [(1300, 2, 30, 1100, torch.float32, 1e-4), (1600, 3, 40, 1500, torch.float32, 1e-4),
                                                 (4200, 1, 28, 4096, torch.float32, 1e-4), (1100, 2, 12, 1025, torch.float64, 1e-9),
                                                 (8300, 1, 20, 8192, torch.float32, 1e-4), (5100, 2, 9, 5000, torch.float64, 1e-9),
                                                 (1200, 2, 200, 1100, torch.float32, 1e-4), (1150, 2, 600, 1030, torch.float32, 1e-4)])
def test_very_long_targets(T, B, N, L, dtype, rtol):
    """1024 < S <= 8192 (the reference has no limit: force_aligned_lattice.cpp:84-154): strip-mined recursion (four / eight positions per
    thread, the frame through LDS) and a frame-by-frame gradient kernel with a fixed-point label row; small, medium and
    resident-slice alphabets; variable lengths, one infeasible utterance when B >= 3; run-to-run determinism."""
    rng = np.random.default_rng(T + L)
    tr, x, tg, _, _ = util.synth(T, B, N, L, L + N)
    il = rng.integers(max(L, T - 60), T + 1, B)
    tl = rng.integers(max(1025, L - 40), L + 1, B)
    tl[0] = L
    il[0] = T
    if B >= 3:
        il[2], tl[2] = 1030, 1100                   # target longer than the input: +inf loss, NaN-free gradients
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, "none")
    r = run_hip(x, tg, tr, il, tl, "none", dtype)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], rtol, "very long targets T%d B%d N%d L%d %s" % (T, B, N, L, k))
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()
    r3 = run_hip(x, tg, tr, il, tl, "none", dtype)
    assert np.array_equal(r["grad_inputs"], r3["grad_inputs"]) and np.array_equal(r["grad_transition"], r3["grad_transition"])
    A = _asg()
    m = A.ASGLoss(N, reduction="none").to(DEV).to(dtype).eval()
    with torch.no_grad():
        m.transition.copy_(tr.to(dtype))
        ev = m(x.to(DEV, dtype), tg.to(DEV), torch.as_tensor(il).to(DEV), torch.as_tensor(tl).to(DEV))
    util.assert_close(ev.cpu().numpy(), o["loss"], rtol, "very long targets, evaluation route")


@pytest.mark.parametrize("shape", [(10, 70, 1300, 4), (8, 100, 2100, 4), (7, 130, 1100, 3)])
def test_streaming_step_on_the_bf16_pipe(shape, monkeypatch):
    """More than 64 utterances over a large alphabet: the per-frame step multiplies on v_mfma_f32_16x16x32_bf16 with every float of the
    matrix and of the vectors as three bfloat16 planes (fwd_step_bf3: six partial products, fp32 accumulation).  Must be fp32-equivalent:
    1e-4 against the fp64 oracle like every route, AND within 2e-6 (scaled) of the same step on the exact fp32 matrix instruction
    (ASG_STEP_NO_BF3=1); 8-nat transitions, variable lengths, an utterance with no alignment, the evaluation route."""
    A = _asg()
    T, B, N, L = shape
    tr, x, tg, il, tl = util.synth(T, B, N, L, 77, True)
    tr = tr * 8.0
    il[1], tl[1] = 2, L                              # target longer than the input: no alignment (+inf loss, finite gradients)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
    res = {}
    for name, env in (("bf3", None), ("fp32", "1")):
        util.setenv(monkeypatch, "ASG_STEP_NO_BF3", env)
        m = A.ASGLoss(N, reduction="none").to(DEV)
        with torch.no_grad():
            m.transition.copy_(tr)
        xd = x.to(DEV).requires_grad_(True)
        loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
        fin = torch.isfinite(loss)
        loss[fin].sum().backward()
        m.eval()
        with torch.no_grad():
            ev = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
        res[name] = (loss.detach().cpu().numpy(), xd.grad.cpu().numpy(), m.transition.grad.cpu().numpy(), ev.cpu().numpy())
    assert np.isinf(o["loss"][1]) and np.isinf(res["bf3"][0][1])
    g = np.isfinite(o["loss"]).astype(np.float64)
    og = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none", grad_out=g)
    for name in ("bf3", "fp32"):
        util.assert_close(res[name][0], o["loss"], 1e-4, name + " loss")
        util.assert_close(res[name][3], o["loss"], 1e-4, name + " evaluation route")
        util.assert_close(res[name][1], og["grad_inputs"], 1e-4, name + " grad_inputs")
        util.assert_close(res[name][2], og["grad_transition"], 1e-4, name + " grad_transition")
    for a, b_, what in zip(res["bf3"], res["fp32"], ("loss", "grad_inputs", "grad_transition", "evaluation")):
        util.assert_close(a, b_, 2e-6, what + ": bf16 pipe vs fp32 matrix instruction")
