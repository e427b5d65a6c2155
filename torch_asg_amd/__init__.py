"""torch_asg_amd -- MI355X (gfx950) native ASG forward-backward hot path behind the ASGLoss surface
of zh217/torch-asg (`from torch_asg import ASGLoss` -> `from torch_asg_amd import ASGLoss`)."""
from .asg import ASGLoss, ASGLossFunction, FAC, FCC, ASGGPUFast, ASGGPUFastForwardOnly, viterbi_align  # noqa: F401
from .distributed import shard_batch, sharded_asg_loss, allreduce_transition_grad  # noqa: F401
from . import native_shim  # noqa: F401  (the reference's `torch_asg_native` on top of libasg_hip.so: native_shim.install())

__all__ = ["ASGLoss", "ASGLossFunction", "FAC", "FCC", "ASGGPUFast", "ASGGPUFastForwardOnly", "viterbi_align",
           "shard_batch", "sharded_asg_loss", "allreduce_transition_grad"]
