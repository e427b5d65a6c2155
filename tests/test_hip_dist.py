"""The RCCL branch on the one GPU a test box has (-m gpu): a one-rank "nccl" process group.

No scaling can be measured on one GPU; what CAN be shown is that the code a multi-GPU run executes -- init_process_group("nccl"),
the all-reduce of transition.grad (captured into the step's hipGraph, or the stated fallback), barrier-fenced timing, the real
HipBackend under shard_batch / sharded_asg_loss -- runs, says what it did, and changes no result.  The world-size-2 arithmetic
is covered on CPU (tests/test_distributed_cpu.py, gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.subprocess_only]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def _bench(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra",
                        "--no-cpu-baseline", "--no-pmc"] + extra, capture_output=True, text=True, timeout=900, env=_env())
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_bench_force_dist_one_rank_rccl():
    plain, _ = _bench([])
    d, err = _bench(["--force-dist"])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 64
    assert d["config"]["collective"] == "rccl all_reduce(transition.grad), 1 rank(s)"
    has = d["config"]["graph_has_collective"]
    assert isinstance(has, bool)
    if has:
        assert "10 consecutive steps per hipGraph replay" in d["config"]["step_mode"]
    else:
        # the stated fallback: the all-reduce could not be captured, one step per replay with the collective after it
        assert "could not capture the all-reduce" in err
        assert "1 consecutive steps per hipGraph replay" in d["config"]["step_mode"]
    # functional facts only (two runs of a 60 us step on a shared box do not agree to a fixed percentage)
    assert d["value"] > 0 and plain["value"] > 0 and d["steps"] == plain["steps"] == 20
    assert plain["config"]["collective"] == "none (one process)" and plain["config"]["graph_has_collective"] is False


def test_hip_backend_under_a_one_rank_nccl_group():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_one_rank.py")], capture_output=True, text=True,
                       timeout=900, env=_env())
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1 and len(d["cases"]) == 3
    for c in d["cases"]:
        assert c["allreduce_bit_identical"], c
        assert all(v <= 1e-4 for v in c["scaled_err"].values()), c
    assert d["big_allreduce_bit_identical"]
    # DistributedDataParallel(ASGLoss) on the real kernels (reduction 'none' inside sharded_asg_loss: the Python Function or the C++ node,
    # whichever the call takes, under DDP's gradient hooks)
    assert d["ddp_grad_fn"] is not None
    assert all(v <= 1e-4 for v in d["ddp_scaled_err"].values()), d["ddp_scaled_err"]
