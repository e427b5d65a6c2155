"""Random shapes through the routes between the fused step and the large-alphabet kernels (long targets over small
alphabets, medium alphabets, both, alphabets of 257 .. 1025 labels (one in five of those draws: 2049 .. 3100), and the boundaries N = 64/65/256/257, S = 64/65/256/257/512/513) against the fp64
oracle: lengths of every kind (infeasible included), all reductions, fp32 and fp64, 30 % of the cases with transition
scores scaled to 5 / 40 nats; ONE gate, 1e-4 (1e-9 in fp64).  tools/fuzz_routes.py is the long form."""
import numpy as np
import pytest
import torch

import util
from oracle import asg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(rng):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        N, S, T = int(rng.integers(2, 65)), int(rng.integers(65, 1025)), int(rng.integers(1, 500))
    elif kind == 1:
        N, S, T = int(rng.integers(65, 257)), int(rng.integers(1, 65)), int(rng.integers(1, 250))
    elif kind == 2:
        N, S, T = int(rng.integers(65, 257)), int(rng.integers(65, 500)), int(rng.integers(1, 300))
    elif kind == 4:
        # 256 < N <= 1024 (matrix resident in a cluster of workgroups) and just beyond (a launch per frame)
        N, S, T = int(rng.choice([257, 300, 448, 512, 513, 640, 777, 1000, 1024, 1025])), int(rng.integers(1, 90)), int(rng.integers(1, 50))
        if rng.random() < 0.2:
            # one draw in five: beyond 2048 labels (fp32: fwd_step_mfma; fp64: fwd_step_kernel<double> + bwd_post_kernel<double, true>)
            N, T = int(rng.choice([2049, 2100, 2600, 3100])), int(rng.integers(1, 14))
    else:
        N = int(rng.choice([64, 65, 128, 129, 192, 193, 256, 257]))
        S = int(rng.choice([64, 65, 128, 129, 256, 257, 512, 513]))
        T = int(rng.integers(2, 160))
    return T, int(rng.integers(1, 4)), N, S


@pytest.mark.parametrize("seed", [3, 17, 29, 41, 53, 67])
def test_random_shapes_between_the_paths(seed):
    import torch_asg_amd
    rng = np.random.default_rng(seed)
    for _ in range(22):
        T, B, N, S = _case(rng)
        dtype = torch.float32 if rng.random() < 0.8 else torch.float64
        tr, x, tg, _, _ = util.synth(T, B, N, S, int(rng.integers(0, 1 << 30)))
        if rng.random() < 0.3:          # transition scores of 5 / 40 nats (trained models; near-forced alignments when tl ~ il)
            tr = tr * float(rng.choice([5.0, 40.0])) - 2.0
        il = rng.integers(1, T + 1, B)
        tl = rng.integers(1, S + 1, B)
        if rng.random() < 0.5:
            il[0] = T
        if rng.random() < 0.5:
            tl[0] = S
        red = ["mean", "sum", "none"][int(rng.integers(0, 3))]
        if (tl > il).any():
            red = "none"                                  # (a reduced loss with an infeasible utterance is +inf)
        m = torch_asg_amd.ASGLoss(N, reduction=red).to(DEV).to(dtype)
        with torch.no_grad():
            m.transition.copy_(tr.to(dtype))
        xd = x.to(DEV, dtype).requires_grad_(True)
        loss = m(xd, tg.to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV))
        fin = torch.isfinite(loss)
        go = fin.cpu().numpy().astype(np.float64) if red == "none" else None
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il, tl, red, grad_out=go)
        tol = 1e-4 if dtype == torch.float32 else 1e-9
        what = "T%d B%d N%d S%d %s %s il=%s tl=%s" % (T, B, N, S, dtype, red, il.tolist(), tl.tolist())
        util.assert_close(loss.detach().cpu().numpy(), o["loss"], tol, what + " loss")
        if fin.any():
            (loss[fin].sum() if red == "none" else loss).backward()
            torch.cuda.synchronize()
            util.assert_close(xd.grad.cpu().numpy(), o["grad_inputs"], tol, what + " grad_inputs")
            util.assert_close(m.transition.grad.cpu().numpy(), o["grad_transition"], tol, what + " grad_transition")
            assert not torch.isnan(xd.grad).any()
