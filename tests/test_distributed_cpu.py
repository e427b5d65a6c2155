"""world_size-2 gloo test of the batch-sharded path: shard along B, run the loss per rank
(oracle-backed stand-in for the HIP binding on CPU), ONE all-reduce(SUM) of transition.grad, and compare with
the unsharded result.  Covers unequal shards and every reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, reduction, B, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch_asg_amd
    import util
    from oracle_backend import OracleBackend
    torch_asg_amd.asg._backend = OracleBackend()
    T, N, L = 12, 6, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True, torch.float64)
    m = torch_asg_amd.ASGLoss(N, reduction=reduction).double()
    with torch.no_grad():
        m.transition.copy_(tr)
    xs, tgs, ils, tls = torch_asg_amd.shard_batch(x, tg, il, tl)          # rank/world from the process group
    xs = xs.clone().requires_grad_(True)
    loss = torch_asg_amd.sharded_asg_loss(m, xs, tgs, ils, tls)
    (loss.sum() if reduction == "none" else loss).backward()
    torch_asg_amd.allreduce_transition_grad(m)
    tot = loss.detach().sum().reshape(1).clone()
    dist.all_reduce(tot)
    lo, hi = torch_asg_amd.distributed.shard_bounds(B, rank, world)
    q.put((rank, lo, hi, float(tot), xs.grad.numpy(), m.transition.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduction,B", [("mean", 6), ("sum", 5), ("none", 4)])
def test_two_rank_sharding_equals_single_process(reduction, B):
    sys.path.insert(0, HERE)
    import util
    from oracle import asg_oracle as orc
    world = 2
    port = 29600 + (os.getpid() % 300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reduction, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T, N, L = 12, 6, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True, torch.float64)
    ref = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), reduction)
    total = np.sum(ref["loss"])
    for rank, lo, hi, tot, gin, gtr in res:
        assert abs(tot - total) < 1e-9 * max(1, abs(total))
        util.assert_close(gtr, ref["grad_transition"], 1e-10, "all-reduced transition.grad")
        util.assert_close(gin, ref["grad_inputs"][:, lo:hi], 1e-10, "local grad_inputs")
    assert sorted((lo, hi) for _, lo, hi, *_ in res) == [(0, (B + 1) // 2), ((B + 1) // 2, B)]


def test_shard_bounds_cover_the_batch():
    from torch_asg_amd.distributed import shard_bounds
    for B in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
