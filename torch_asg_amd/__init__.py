"""torch_asg_amd -- MI355X (gfx950) native ASG forward-backward hot path behind the ASGLoss surface
of zh217/torch-asg (`from torch_asg import ASGLoss` -> `from torch_asg_amd import ASGLoss`)."""
from .asg import ASGLoss, ASGLossFunction, FAC, FCC, ASGGPUFast, ASGGPUFastForwardOnly, viterbi_align  # noqa: F401
from .distributed import shard_batch, sharded_asg_loss, allreduce_transition_grad  # noqa: F401
from ._graphed import graphed, GraphedStep  # noqa: F401
from . import native_shim  # noqa: F401  (the reference's `torch_asg_native` on top of libasg_hip.so: native_shim.install())


def reserve(device, nbytes=0):
    """Create the zeroed sync pool of `device` now (the fused step needs one; it cannot be created while a hipGraph is being
    captured -- call this, or run one warm-up step, before the capture)."""
    from .asg import native
    return native().reserve(device, nbytes)


def check_faults():
    """Raise RuntimeError if a resident-slice forward launch (256 < N <= 2048 in fp32, <= 1024 in fp64) of this process timed out since the last look --
    the step that contained it returned NaN scores and gradients and must be discarded (the library has already switched to kernels
    that need no co-residency, so repeating the step is sound).  ASGLoss checks at the start of every forward AND backward call with
    N > 256; a training loop that wants to know before `optimizer.step()` calls this after its own synchronisation point
    (`loss.item()`): it reads one host-pinned word, no device synchronisation of its own."""
    from .asg import native
    return native().check_faults()


def release():
    """Destroy the side-stream contexts and drop the sync pools (call when no launch of this process is in flight)."""
    from .asg import native
    return native().release()


__all__ = ["ASGLoss", "ASGLossFunction", "FAC", "FCC", "ASGGPUFast", "ASGGPUFastForwardOnly", "viterbi_align",
           "shard_batch", "sharded_asg_loss", "allreduce_transition_grad", "reserve", "release", "check_faults", "graphed", "GraphedStep"]
