"""CPU tests: pin oracle/asg_oracle.c (the checker) to the reference.

(a) known-answer vectors of the reference's own tests (re-typed data + expected values),
(b) committed golden fixtures generated from the real reference (tests/golden/make_golden.py),
(c) live comparison with the reference's compiled CPU path (oracle/_ref) when it is present,
(d) brute-force path enumeration (independent of any recursion).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc
from oracle import ref_runner


# ---------------------------------------------------------------- (a) known answers
def test_fac_1_known_answer():
    # /root/reference/torch_asg/test/test_asg.py:189-224
    B, T, S, N = 2, 3, 2, 2
    x = np.array([1.0, 0.0, 0.0, 1.0, 0.5, 0.5, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0]).reshape(B, T, N).transpose(1, 0, 2)
    tr = np.zeros((N, N))
    tg = np.array([[0, 1], [0, 1]])
    s, _, _ = orc.aligned_forward(x, tg, tr, [T, T], [S, S])
    exp = np.array([np.logaddexp(1.5, 2.5), np.logaddexp(2.0, 3.0)])
    assert np.abs(s - exp).sum() < 1e-10


def test_fac_2_known_answer():
    # test_asg.py:227-254: -log 32
    B, T, S, N = 1, 3, 2, 4
    x = np.full((T, B, N), math.log(0.25))
    s, _, _ = orc.aligned_forward(x, np.array([[0, 1]]), np.zeros((N, N)), [T], [S])
    assert abs(s[0] + math.log(32.0)) < 1e-10


@pytest.mark.parametrize("dtype,eps", [(np.float64, 1e-10), (np.float32, 1e-4)])
def test_fcc_normalised_invariant(dtype, eps):
    # test_asg.py:49-128: normalised emissions + zero transitions => S_full = 0
    rng = np.random.default_rng(0)
    B, T, N = 3, 300, 40
    p = rng.uniform(size=(B, T, N))
    x = np.log(p / p.sum(-1, keepdims=True)).transpose(1, 0, 2).astype(dtype)
    s, _, _ = orc.full_forward(x, np.zeros((N, N), dtype), [T] * B)
    assert np.abs(s).sum() < eps


def test_asg_1_and_2_known_answers():
    # test_asg.py:291-321  expected [log 2, 0];  :324-351 expected log 32
    B, T, S, N = 2, 3, 2, 2
    with np.errstate(divide="ignore"):
        x = np.log(np.array([1.0, 0.0, 0.0, 1.0, 0.5, 0.5, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0])
                   .reshape(B, T, N).transpose(1, 0, 2))
    r = orc.asg_loss(x, np.array([[0, 1], [0, 1]]), np.zeros((N, N)), [T] * B, [S] * B, "none")
    assert np.abs(r["loss"] - np.array([math.log(2.0), 0.0])).sum() < 1e-10
    x = np.full((3, 1, 4), math.log(0.25))
    r = orc.asg_loss(x, np.array([[0, 1]]), np.zeros((4, 4)), [3], [2], "mean")
    assert abs(r["loss"] - math.log(32.0)) < 1e-10


def _asg4():
    # test_asg.py:379-464 (wav2letter-style vectors): data + expected values
    B, T, S, N = 3, 5, 5, 6
    x = np.array([
        -0.4340, -0.0254, +0.3667, +0.4180, -0.3805, -0.1707, +0.1060, +0.3631, -0.1122, -0.3825, -0.0031, -0.3801,
        +0.0443, -0.3795, +0.3194, -0.3130, +0.0094, +0.1560, +0.1252, +0.2877, +0.1997, -0.4554, +0.2774, -0.2526,
        -0.4001, -0.2402, +0.1295, +0.0172, +0.1805, -0.3299,
        +0.3298, -0.2259, -0.0959, +0.4909, +0.2996, -0.2543, -0.2863, +0.3239, -0.3988, +0.0732, -0.2107, -0.4739,
        -0.0906, +0.0480, -0.1301, +0.3975, -0.3317, -0.1967, +0.4372, -0.2006, +0.0094, +0.3281, +0.1873, -0.2945,
        +0.2399, +0.0320, -0.3768, -0.2849, -0.2248, +0.3186,
        +0.0225, -0.3867, -0.1929, -0.2904, -0.4958, -0.2533, +0.4001, -0.1517, -0.2799, -0.2915, +0.4198, +0.4506,
        +0.1446, -0.4753, -0.0711, +0.2876, -0.1851, -0.1066, +0.2081, -0.1190, -0.3902, -0.1668, +0.1911, -0.2848,
        -0.3846, +0.1175, +0.1052, +0.2172, -0.0362, +0.3055]).reshape(B, T, N)
    tg = np.array([2, 1, 5, 1, 3, 4, 3, 5, 0, 0, 3, 2, 2, 1, 0]).reshape(B, S)
    loss = np.array([7.7417464256287, 6.4200420379639, 8.2780694961548])
    gi = np.array([
        0.1060, 0.1595, -0.7639, 0.2485, 0.1118, 0.1380, 0.1915, -0.7524, 0.1539, 0.1175, 0.1717, 0.1178,
        0.1738, 0.1137, 0.2288, 0.1216, 0.1678, -0.8057, 0.1766, -0.7923, 0.1902, 0.0988, 0.2056, 0.1210,
        0.1212, 0.1422, 0.2059, -0.8160, 0.2166, 0.1300,
        0.2029, 0.1164, 0.1325, 0.2383, -0.8032, 0.1131, 0.1414, 0.2602, 0.1263, -0.3441, -0.3009, 0.1172,
        0.1557, 0.1788, 0.1496, -0.5498, 0.0140, 0.0516, 0.2306, 0.1219, 0.1503, -0.4244, 0.1796, -0.2579,
        0.2149, 0.1745, 0.1160, 0.1271, 0.1350, -0.7675,
        0.2195, 0.1458, 0.1770, -0.8395, 0.1307, 0.1666, 0.2148, 0.1237, -0.6613, -0.1223, 0.2191, 0.2259,
        0.2002, 0.1077, -0.8386, 0.2310, 0.1440, 0.1557, 0.2197, -0.1466, -0.5742, 0.1510, 0.2160, 0.1342,
        0.1050, -0.8265, 0.1714, 0.1917, 0.1488, 0.2094]).reshape(B, T, N)
    gt = np.array([
        0.3990, 0.3396, 0.3486, 0.3922, 0.3504, 0.3155, 0.3666, 0.0116, -1.6678, 0.3737, 0.3361, -0.7152,
        0.3468, 0.3163, -1.1583, -0.6803, 0.3216, 0.2722, 0.3694, -0.6688, 0.3047, -0.8531, -0.6571, 0.2870,
        0.3866, 0.3321, 0.3447, 0.3664, -0.2163, 0.3039, 0.3640, -0.6943, 0.2988, -0.6722, 0.3215, -0.1860]).reshape(N, N)
    return x, tg, np.array([T] * B), np.array([5, 3, 4]), loss, gi, gt


ASG4 = _asg4


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_asg_4_known_answer(dtype):
    xb, tg, il, tl, loss, gi, gt = _asg4()
    x = xb.astype(dtype).transpose(1, 0, 2)        # non-contiguous permuted view, as test_asg.py:454
    assert not x.flags["C_CONTIGUOUS"]
    r = orc.asg_loss(x, tg, np.zeros((6, 6), dtype), il, tl, "none")
    assert np.abs(r["loss"] - loss).sum() < 1e-3
    assert np.abs(r["grad_inputs"].transpose(1, 0, 2) - gi).max() < 1e-4
    assert np.abs(r["grad_transition"] - gt).max() < 1e-4


# ---------------------------------------------------------------- (b) golden fixtures
def _run_small(g, dtype):
    kw = {}
    il, tl = g["input_lengths"], g["target_lengths"]
    if not bool(g["pass_lengths"]):
        il = tl = None
    return orc.asg_loss(g["inputs"].astype(dtype), g["targets"], g["transition"].astype(dtype), il, tl,
                        str(g["reduction"]))


@pytest.mark.parametrize("name", util.SMALL)
@pytest.mark.parametrize("tag,dtype,rtol", [("f64", np.float64, 1e-9), ("f32", np.float32, 1e-4)])
def test_golden_small(name, tag, dtype, rtol):
    g = util.load(name)
    r = _run_small(g, dtype)
    for k in ("loss", "full_scores", "aligned_scores", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], g["%s_%s" % (tag, k)], rtol, "%s/%s/%s" % (name, tag, k))
    assert not np.isnan(r["grad_inputs"]).any() and not np.isnan(r["grad_transition"]).any()


@pytest.mark.parametrize("name", ["cfg2", "cfg2_var", "cfg3_var"])
def test_golden_large_f64(name):
    g = util.load(name)
    tr, x, tg, il, tl = util.synth(int(g["T"]), int(g["B"]), int(g["N"]), int(g["L"]), int(g["seed"]),
                                   bool(g["variable"]), torch.float64)
    assert abs(float(x.sum()) - float(g["inputs_checksum"])) < 1e-6
    r = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), str(g["reduction"]))
    util.assert_close(r["loss"], g["f64_loss"], 1e-10, name + "/loss")
    util.assert_close(r["grad_transition"], g["f64_grad_transition"], 1e-9, name + "/gtr")
    util.assert_close(r["grad_inputs"][::7, ::3, :], g["f64_grad_inputs_sample"], 1e-9, name + "/gin")
    util.assert_close(r["grad_inputs"].sum(0), g["f64_grad_inputs_sum_t"], 1e-9, name + "/gin_sum")


def test_golden_cfg2_f32():
    g = util.load("cfg2_var")
    tr, x, tg, il, tl = util.synth(150, 16, 30, 20, 0, True, torch.float32)
    r = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "mean")
    util.assert_close(r["loss"], g["f32_loss"], 1e-4, "loss")
    util.assert_close(r["grad_transition"], g["f32_grad_transition"], 1e-4, "gtr")
    util.assert_close(r["grad_inputs"][::7, ::3, :], g["f32_grad_inputs_sample"], 1e-4, "gin")


# ---------------------------------------------------------------- (c) live reference
@pytest.mark.skipif(not ref_runner.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_live_reference_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    T, B, N, L = int(rng.integers(2, 30)), int(rng.integers(1, 5)), int(rng.integers(2, 12)), int(rng.integers(1, 8))
    tr, x, tg, _, _ = util.synth(T, B, N, L, seed, False, torch.float64)
    il = torch.from_numpy(rng.integers(1, T + 1, B))
    tl = torch.from_numpy(rng.integers(1, L + 1, B))
    ref = ref_runner.asg_loss(x, tg, tr, il, tl, "sum")
    r = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "sum")
    for k in ("loss_per_utt", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], ref[k].numpy(), 1e-10, k)


# ---------------------------------------------------------------- (d) brute force
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_brute_force_enumeration(seed):
    rng = np.random.default_rng(seed)
    T, N, L = 5, 3, 3
    x = rng.normal(size=(T, 1, N))
    tr = rng.normal(size=(N, N))
    tg = rng.integers(0, N, (1, L))
    il, tl = int(rng.integers(3, T + 1)), int(rng.integers(1, L + 1))
    f, _, _ = orc.full_forward(x, tr, [il])
    a, _, _ = orc.aligned_forward(x, tg, tr, [il], [tl])
    bf, ba = orc.brute_force_scores(x[:, 0, :], tg[0], tr, il, tl)
    assert abs(f[0] - bf) < 1e-10 and abs(a[0] - ba) < 1e-10


@pytest.mark.parametrize("seed", range(6))
def test_viterbi_oracle_against_exhaustive_enumeration(seed):
    """The reference has no best-path alignment (README.md:33 TODO): the oracle's max-plus restatement of
    force_aligned_lattice.cpp:84-111 is pinned by enumerating every alignment of tiny utterances."""
    rng = np.random.default_rng(100 + seed)
    for _ in range(40):
        T, N, L = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
        x = rng.normal(size=(T, 1, N))
        tr = rng.normal(size=(N, N))
        tg = rng.integers(0, N, size=(1, L))
        il, tl = np.array([rng.integers(1, T + 1)]), np.array([rng.integers(1, L + 1)])
        sc, path = orc.viterbi(x, tg, tr, il, tl)
        best, bp = orc.brute_force_viterbi(x[:, 0], tg[0], tr, il[0], tl[0])
        if bp is None:
            assert sc[0] == -np.inf and (path == -1).all()
        else:
            assert abs(sc[0] - best) < 1e-12
            assert list(path[0, :il[0]]) == bp and (path[0, il[0]:] == -1).all()
            # the best path can never beat the sum over all alignments, and is within log(#paths) of it
            ali = orc.aligned_forward(x, tg, tr, il, tl)[0][0]
            assert sc[0] <= ali + 1e-12


def test_viterbi_oracle_f32_and_ties():
    # all-zero scores: every alignment ties; "stay" wins every tied comparison, so walking the back-pointers from the
    # end stays on a position for as long as it was reachable: the best path advances as EARLY as possible
    x = np.zeros((6, 1, 3), np.float32)
    sc, path = orc.viterbi(x, np.array([[0, 1, 2]]), np.zeros((3, 3), np.float32), None, None)
    assert sc[0] == 0 and list(path[0]) == [0, 1, 2, 2, 2, 2]
