"""Batch-sharded multi-GPU use of the ASG hot path (not present in the reference).

Utterances are independent, so the batch dimension shards across ranks with no data-path
collective: each rank runs the forward/backward kernels on its B/world utterances.  The only
shared quantity is the transition matrix (replicated, read-only in forward) and its gradient
grad_Tr = sum_b (...), which needs ONE all-reduce(SUM) per step (RCCL over xGMI when the process
group backend is "nccl"; "gloo" in the CPU tests).

These helpers are engine-agnostic: they only slice tensors and call torch.distributed.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, rank, world):
    """Contiguous, balanced [lo, hi) slice of the batch for `rank` (first `batch % world` ranks get one more)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(inputs, targets, input_lengths=None, target_lengths=None, rank=None, world=None):
    """Slice a global batch [T,B,N] / [B,S] / [B] / [B] along B for this rank."""
    if rank is None:
        rank = dist.get_rank()
    if world is None:
        world = dist.get_world_size()
    lo, hi = shard_bounds(inputs.shape[1], rank, world)
    il = None if input_lengths is None else input_lengths[lo:hi]
    tl = None if target_lengths is None else target_lengths[lo:hi]
    return inputs[:, lo:hi], targets[lo:hi], il, tl


def _unwrap(loss_module):
    return getattr(loss_module, "module", loss_module) if not hasattr(loss_module, "transition") else loss_module


def sharded_asg_loss(loss_module, inputs, targets, input_lengths=None, target_lengths=None,
                     global_batch=None, reduction=None):
    """Loss of this rank's shard, scaled so that summing over ranks gives the global-batch loss.

    Two ways to the same `transition.grad`: call `allreduce_transition_grad` after backward (a SUM: the
    gradient of the global-batch loss on every rank), or wrap the module in `DistributedDataParallel` and pass the
    wrapper here -- DDP then all-reduces and AVERAGES, so every rank holds 1/world of it
    (tests/test_distributed_cpu.py::test_ddp_wrapped_module_averages_what_the_sum_route_adds).

    `loss_module` must be an ASGLoss-like module; it is evaluated with reduction 'none' and reduced here:
    'mean' divides by the GLOBAL batch size so local gradients are already correctly scaled and the
    all-reduce of transition.grad is a plain SUM (no post-division, exact for unequal shards too).
    """
    inner = _unwrap(loss_module)            # DistributedDataParallel(ASGLoss): the settings live on .module, the call goes through the wrapper
    reduction = reduction or inner.reduction
    old = inner.reduction
    inner.reduction = 'none'
    try:
        per_utt = loss_module(inputs, targets, input_lengths, target_lengths)
    finally:
        inner.reduction = old
    if reduction == 'none':
        return per_utt
    total = per_utt.sum()
    if reduction == 'mean':
        if global_batch is None:
            n = torch.tensor([per_utt.numel()], dtype=torch.int64, device=per_utt.device)
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(n)
            global_batch = int(n.item())
        total = total / global_batch
    return total


def allreduce_transition_grad(loss_module, group=None, async_op=False, force=False):
    """The single collective of the step: all-reduce(SUM) of transition.grad across ranks.
    A one-rank group has nothing to add and is skipped; `force=True` issues the collective anyway (tests that want the
    RCCL call itself to run on a one-GPU box)."""
    g = _unwrap(loss_module).transition.grad
    if g is None:
        raise RuntimeError("transition.grad is None: call backward() first")
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size(group) == 1 and not force:
        return None
    return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
