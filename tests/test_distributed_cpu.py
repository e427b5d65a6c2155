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


def _spawn(target, world, extra, timeout=600):
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(extra) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("reduction,B", [("mean", 512), ("mean", 13), ("sum", 13)])
def test_eight_rank_sharding_equals_single_process(reduction, B):
    """The node's shape (BASELINE.json configs[3]): 8 ranks; B = 512 -> 64 utterances per rank (cfg 4), and B = 13 ->
    shards of 2, 2, 2, 2, 2, 1, 1, 1.  One all-reduce(SUM) of transition.grad; every rank ends with the single-process gradient."""
    sys.path.insert(0, HERE)
    import util
    from oracle import asg_oracle as orc
    world = 8
    res = _spawn(_worker, world, (reduction, B))
    T, N, L = 12, 6, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True, torch.float64)
    ref = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), reduction)
    total = np.sum(ref["loss"])
    from torch_asg_amd.distributed import shard_bounds
    for rank, lo, hi, tot, gin, gtr in res:
        assert (lo, hi) == shard_bounds(B, rank, world)
        assert abs(tot - total) < 1e-9 * max(1, abs(total))
        util.assert_close(gtr, ref["grad_transition"], 1e-10, "all-reduced transition.grad, rank %d" % rank)
        util.assert_close(gin, ref["grad_inputs"][:, lo:hi], 1e-10, "local grad_inputs, rank %d" % rank)
    sizes = sorted(hi - lo for _, lo, hi, *_ in res)
    assert sizes == ([64] * 8 if B == 512 else [1, 1, 1, 2, 2, 2, 2, 2])


def _ddp_worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch_asg_amd
    import util
    from oracle_backend import OracleBackend
    from torch.nn.parallel import DistributedDataParallel as DDP
    torch_asg_amd.asg._backend = OracleBackend()
    T, N, L = 12, 6, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True, torch.float64)
    m = torch_asg_amd.ASGLoss(N, reduction="mean").double()
    with torch.no_grad():
        m.transition.copy_(tr)
    ddp = DDP(m)                                                          # transition is a Parameter: DDP all-reduces its gradient
    xs, tgs, ils, tls = torch_asg_amd.shard_batch(x, tg, il, tl)
    xs = xs.clone().requires_grad_(True)
    loss = torch_asg_amd.sharded_asg_loss(ddp, xs, tgs, ils, tls, global_batch=B)
    loss.backward()                                                       # the all-reduce (AVERAGE) happens inside, overlapped by DDP
    lo, hi = torch_asg_amd.distributed.shard_bounds(B, rank, world)
    q.put((rank, lo, hi, float(loss), xs.grad.numpy(), m.transition.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_wrapped_module_averages_what_the_sum_route_adds():
    """SURVEY.md 8(e), last sentence: wrapping ASGLoss in DistributedDataParallel gives the collective for free because `transition` is a
    Parameter.  DDP averages: every rank's transition.grad x world equals what `allreduce_transition_grad` (SUM) leaves -- the gradient of
    the global-batch loss; the local grad_inputs are untouched by the wrapper."""
    sys.path.insert(0, HERE)
    import util
    from oracle import asg_oracle as orc
    world, B = 4, 10                                                      # shards of 3, 3, 2, 2
    res = _spawn(_ddp_worker, world, (B,))
    T, N, L = 12, 6, 4
    tr, x, tg, il, tl = util.synth(T, B, N, L, 5, True, torch.float64)
    ref = orc.asg_loss(x.numpy(), tg.numpy(), tr.numpy(), il.numpy(), tl.numpy(), "mean")
    assert abs(sum(r[3] for r in res) - float(ref["loss"])) < 1e-9 * max(1, abs(float(ref["loss"])))
    for rank, lo, hi, _, gin, gtr in res:
        util.assert_close(gtr * world, ref["grad_transition"], 1e-10, "DDP-averaged transition.grad x world, rank %d" % rank)
        util.assert_close(gin, ref["grad_inputs"][:, lo:hi], 1e-10, "local grad_inputs, rank %d" % rank)


def test_bench_eight_ranks_through_the_real_launcher_dry_run():
    """`bench.py --gpus 8` exactly as the driver starts it (torch.distributed.run, 8 ranks, rendezvous on 127.0.0.1), on gloo with
    --dry-run: every rank reaches a collective, rank 0 prints ONE JSON line that names 8 ranks and the cfg-4 global batch."""
    import json
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20",
                          "--warmup", "5", "--dry-run"], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world"] == 8 and d["ranks_joined"] == 8 and d["global_batch"] == 512
    assert d["collective"] == "rccl all_reduce(transition.grad), 8 rank(s)" and d["uses_dist"] and d["steps_per_graph"] == 10
