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
    """Number of resident-slice forward launches (256 < N <= 2048 in fp32, <= 1024 in fp64) of this process that timed out since the
    last look (a RuntimeWarning says so).  Results are NOT affected: the call that contained the launch repaired itself in stream
    (`fwd_repair_kernel`, exact, tens of milliseconds) before its scores were read, and later calls take kernels that need no
    co-residency (2-3x slower).  ASGLoss looks at the start of every forward call with N > 256; reads one host-pinned word, no
    device synchronisation."""
    from .asg import native
    return native().check_faults()


def release():
    """Destroy the side-stream contexts and drop the sync pools (call when no launch of this process is in flight)."""
    from .asg import native
    return native().release()


__all__ = ["ASGLoss", "ASGLossFunction", "FAC", "FCC", "ASGGPUFast", "ASGGPUFastForwardOnly", "viterbi_align",
           "shard_batch", "sharded_asg_loss", "allreduce_transition_grad", "reserve", "release", "check_faults", "graphed", "GraphedStep"]
