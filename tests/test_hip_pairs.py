"""Large-batch routes of the small-alphabet path, forced on at small sizes (-m gpu): ASG_PAIR_MIN_B=1 -- the aligned recursions two
utterances per wavefront (aligned_pair_chain, default from B = 2048) and the aligned-only kernel.
Against the fp64 oracle AND against the one-utterance chains on the same inputs: variable lengths (one-frame utterances included), every
alphabet tile, label counts that are not multiples of 4 (scalar accesses), -inf emissions, transition scores of tens of nats (flagged
utterances -> exact redo), batch-major strided emissions, evaluation route, reductions, launch modes.
(Round 4's opt-in matrix-core forward for sixteen utterances per workgroup lost its measurement and lives in tools/experiments/.)"""
import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(x, tg, tr, il, tl, red, eval_route=False, mode="single"):
    import torch_asg_amd
    N = tr.shape[0]
    m = torch_asg_amd.ASGLoss(N, reduction=red, launch_mode=mode).to(DEV)
    with torch.no_grad():
        m.transition.copy_(tr)
    xd = x.to(DEV).requires_grad_(True)
    if eval_route:
        m.eval()
        with torch.no_grad():
            return dict(loss=m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV)).cpu().numpy())
    loss = m(xd, tg.to(DEV), il.to(DEV), tl.to(DEV))
    fin = torch.isfinite(loss)
    (loss[fin].sum() if red == "none" else loss).backward()
    torch.cuda.synchronize()
    return dict(loss=loss.detach().cpu().numpy(), grad_inputs=xd.grad.cpu().numpy(), grad_transition=m.transition.grad.cpu().numpy())


SHAPES = [(50, 16, 40, 10), (130, 37, 40, 30), (1, 5, 40, 1), (2, 20, 40, 2), (9, 33, 30, 5), (60, 100, 48, 20), (40, 17, 64, 12),
          (33, 48, 8, 4), (70, 19, 33, 9), (45, 64, 44, 11), (25, 40, 16, 6), (64, 31, 56, 9), (55, 18, 63, 8), (400, 64, 40, 30)]


@pytest.mark.parametrize("variant", ["plain", "scaled", "neginf", "strided"])
def test_pair_chains_against_oracle_and_per_utterance_chains(variant, monkeypatch):
    rng = np.random.default_rng(len(variant))
    for (T, B, N, L) in SHAPES:
        tr, x, tg, _, _ = util.synth(T, B, N, L, int(rng.integers(0, 1 << 30)))
        il = torch.from_numpy(rng.integers(1, T + 1, B))
        tl = torch.from_numpy(rng.integers(1, L + 1, B))
        if rng.random() < 0.5:
            il[0] = T
        if rng.random() < 0.3:
            il[:] = T
        if variant == "scaled":                     # row sums leave 2^+-100: flagged utterances, exact redo
            tr = tr * 30.0 - 10.0
            x = x * 4.0 - 30.0
        if variant == "neginf":
            x[:, :, 1] = float("-inf")
            tg = torch.where(tg == 1, torch.zeros_like(tg), tg)
            tr[2, :] = -300.0
        if variant == "strided":                    # batch-major view with a padded label axis
            x = torch.randn(B, T, N + 3, generator=torch.Generator().manual_seed(T + B))[:, :, :N].transpose(0, 1)
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none")
        fin = np.isfinite(o["loss"])
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "none", grad_out=fin.astype(np.float64))
        util.setenv(monkeypatch, "ASG_PAIR_MIN_B", "1")
        ra = _run(x, tg, tr, il, tl, "none")
        ev = _run(x, tg, tr, il, tl, "none", eval_route=True)
        util.setenv(monkeypatch, "ASG_PAIR_MIN_B", str(1 << 30))
        rb = _run(x, tg, tr, il, tl, "none")
        what = "T%d B%d N%d L%d %s" % (T, B, N, L, variant)
        for k in ("loss", "grad_inputs", "grad_transition"):
            util.assert_close(ra[k], o[k], 1e-4, what + " vs oracle: " + k)
            util.assert_close(ra[k], rb[k], 2e-5, what + " vs per-utterance chains: " + k)
            assert not np.isnan(ra[k][np.isfinite(o[k])]).any()
        util.assert_close(ev["loss"], o["loss"], 1e-4, what + " evaluation route")


@pytest.mark.parametrize("mode", ["single", "streams", "serial"])
@pytest.mark.parametrize("red", ["mean", "sum"])
def test_pair_routes_reduced_losses_launch_modes_determinism(mode, red, monkeypatch):
    util.setenv(monkeypatch, "ASG_PAIR_MIN_B", "1")
    T, B, N, L = 80, 70, 40, 12
    tr, x, tg, il, tl = util.synth(T, B, N, L, 3, True)
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
    r = _run(x, tg, tr, il, tl, red, mode=mode)
    r2 = _run(x, tg, tr, il, tl, red, mode=mode)
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(r[k], o[k], 1e-4, "%s %s %s" % (red, mode, k))
        assert np.array_equal(r[k], r2[k]), "run-to-run: %s" % k


def test_pair_chains_are_the_default_at_2048_utterances_and_match_the_single_chains(monkeypatch):
    """T small, B = 2048: the default route takes two utterances per aligned wavefront; equal to the one-utterance chains to rounding,
    both within 1e-4 of the oracle on a sample of utterances."""
    T, B, N, L = 24, 2048, 40, 9
    tr, x, tg, il, tl = util.synth(T, B, N, L, 9, True)
    util.setenv(monkeypatch, "ASG_PAIR_MIN_B", None)
    ra = _run(x, tg, tr, il, tl, "none")
    util.setenv(monkeypatch, "ASG_PAIR_MIN_B", str(1 << 30))
    rb = _run(x, tg, tr, il, tl, "none")
    for k in ("loss", "grad_inputs", "grad_transition"):
        util.assert_close(ra[k], rb[k], 2e-5, "pairs vs singles: " + k)
    sel = slice(0, 64)
    o = orc.asg_loss(x[:, sel].double().numpy(), tg[sel].numpy(), tr.double().numpy(), il[sel].numpy(), tl[sel].numpy(), "none")
    util.assert_close(ra["loss"][sel], o["loss"], 1e-4, "pairs vs oracle: loss")
    util.assert_close(ra["grad_inputs"][:, sel], o["grad_inputs"], 1e-4, "pairs vs oracle: grad_inputs")
