"""torch_asg_amd -- MI355X (gfx950) native ASG forward-backward hot path behind the ASGLoss surface
of zh217/torch-asg (`from torch_asg import ASGLoss` -> `from torch_asg_amd import ASGLoss`)."""
from .asg import ASGLoss, ASGLossFunction, FAC, FCC, ASGGPUFast, ASGGPUFastForwardOnly, viterbi_align  # noqa: F401
from .distributed import shard_batch, sharded_asg_loss, allreduce_transition_grad  # noqa: F401
from . import native_shim  # noqa: F401  (the reference's `torch_asg_native` on top of libasg_hip.so: native_shim.install())


def reserve(device, nbytes=0):
    """Create the zeroed sync pool of `device` now (the fused step needs one; it cannot be created while a hipGraph is being
    captured -- call this, or run one warm-up step, before the capture)."""
    from .asg import native
    return native().reserve(device, nbytes)


def release():
    """Destroy the side-stream contexts and drop the sync pools (call when no launch of this process is in flight)."""
    from .asg import native
    return native().release()


__all__ = ["ASGLoss", "ASGLossFunction", "FAC", "FCC", "ASGGPUFast", "ASGGPUFastForwardOnly", "viterbi_align",
           "shard_batch", "sharded_asg_loss", "allreduce_transition_grad", "reserve", "release"]
