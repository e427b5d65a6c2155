"""Worker of tests/test_hip_dist.py: ONE rank, backend "nccl" (= RCCL), on cuda:0.

The real HipBackend (no oracle-backed stand-in) under an initialised process group: shard_batch + sharded_asg_loss +
allreduce_transition_grad(force=True) -- the calls bench.py --gpus N and a training loop make -- checked against the fp64
oracle here, and the all-reduce of a one-rank group must leave transition.grad bit-identical.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import util
    import torch_asg_amd
    from oracle import asg_oracle as orc
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": []}
    for (T, B, N, L, red) in [(150, 16, 30, 20, "mean"), (40, 5, 9, 6, "sum"), (30, 7, 100, 8, "none")]:
        tr, x, tg, il, tl = util.synth(T, B, N, L, 3, True)
        m = torch_asg_amd.ASGLoss(N, reduction=red).to(dev)
        assert type(torch_asg_amd.asg.native()).__name__ == "HipBackend"
        with torch.no_grad():
            m.transition.copy_(tr)
        xs, tgs, ils, tls = torch_asg_amd.shard_batch(x.to(dev), tg.to(dev), il.to(dev), tl.to(dev))   # rank / world from the group
        xs = xs.clone().requires_grad_(True)
        loss = torch_asg_amd.sharded_asg_loss(m, xs, tgs, ils, tls)
        (loss.sum() if red == "none" else loss).backward()
        torch.cuda.synchronize()
        before = m.transition.grad.clone()
        torch_asg_amd.allreduce_transition_grad(m, force=True)                 # the RCCL call itself
        torch.cuda.synchronize()
        o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), red)
        ok = {}
        for k, v in (("loss", loss.detach().cpu().numpy()), ("grad_inputs", xs.grad.cpu().numpy()),
                     ("grad_transition", m.transition.grad.cpu().numpy())):
            ok[k] = util.tol_ok(v, o[k], 1e-4)[1]
        out["cases"].append({"shape": [T, B, N, L], "reduction": red, "scaled_err": ok,
                             "allreduce_bit_identical": bool(torch.equal(before, m.transition.grad))})
    # DistributedDataParallel around the module (`transition` is a Parameter, DDP all-reduces its gradient): the C++ autograd
    # node under DDP's hooks on the real kernels, one rank, nccl -- the averaged gradient of a one-rank group is the gradient
    from torch.nn.parallel import DistributedDataParallel as DDP
    T, B, N, L = 150, 16, 30, 20
    tr, x, tg, il, tl = util.synth(T, B, N, L, 4, True)
    m = torch_asg_amd.ASGLoss(N, reduction="mean").to(dev)
    with torch.no_grad():
        m.transition.copy_(tr)
    ddp = DDP(m, device_ids=[0])
    xd = x.to(dev).requires_grad_(True)
    loss = torch_asg_amd.sharded_asg_loss(ddp, xd, tg.to(dev), il.to(dev), tl.to(dev), global_batch=B)
    out["ddp_grad_fn"] = loss.grad_fn.name() if loss.grad_fn is not None else None
    loss.backward()
    torch.cuda.synchronize()
    o = orc.asg_loss(x.double().numpy(), tg.numpy(), tr.double().numpy(), il.numpy(), tl.numpy(), "mean")
    out["ddp_scaled_err"] = {"loss": util.tol_ok(loss.detach().cpu().numpy(), o["loss"], 1e-4)[1],
                             "grad_inputs": util.tol_ok(xd.grad.cpu().numpy(), o["grad_inputs"], 1e-4)[1],
                             "grad_transition": util.tol_ok(m.transition.grad.cpu().numpy(), o["grad_transition"], 1e-4)[1]}
    # the collective on a tensor of cfg 5's gradient size (400 MB) too: one rank, so still the identity
    big = torch.randn(10000, 10000, device=dev)
    ref = big.clone()
    dist.all_reduce(big)
    torch.cuda.synchronize()
    out["big_allreduce_bit_identical"] = bool(torch.equal(big, ref))
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
