"""`torch_asg_native` as the reference's Python layer imports it, backed by libasg_hip.so.

The reference's `torch_asg/asg.py` does `import torch_asg_native` (asg.py:4) and calls the seven functions that
`native/extension.cpp:15-29` exports.  This module exports the same seven names with the same positional arguments and
the same number of results, so the reference's `asg.py` runs UNMODIFIED on the HIP kernels:

    import torch_asg_amd.native_shim as shim
    shim.install()                       # sys.modules["torch_asg_native"] = this module
    from torch_asg import ASGLoss        # the reference's own package, its asg.py untouched

What differs from the pybind module, and why it does not matter to asg.py:
  * the reference returns alpha / beta / gamma / path_contrib tensors ([T,B,N], [T,B,S], [T-1,B,N,N] ...: 169 MB at
    cfg 3, 25.6 TB at cfg 5) which asg.py only saves, reads `.shape` of (asg.py:28-29, 50, 87-88) and hands back.  The
    HIP path keeps ONE opaque state buffer and recomputes instead of reading path_contrib, so the tensors returned
    here are zero-stride HANDLES of the right shape (no memory behind them); the first of them carries the state
    buffer and the tensors the backward pass needs (inputs, transition, targets, lengths -- asg.py does not save
    inputs on these routes) as a Python attribute.  PyTorch hands the same Python object back from
    `ctx.saved_tensors` (a tensor's Python object and its attributes live as long as the tensor does); a small
    registry keyed by the handle's address is the fallback.
  * CPU tensors raise (this package has no CPU implementation; the reference would run its OpenMP path).
"""
import collections
import sys
import threading

import torch

from . import _lib
from .asg import native

__all__ = ["fully_connected_forward", "fully_connected_backward", "force_aligned_forward", "force_aligned_backward",
           "fast_asg_gpu_forward_only", "fast_asg_gpu_forward", "fast_asg_gpu_backward", "install", "uninstall"]

_MAX_PENDING = 4         # fallback registry only: bounds what a forward without a backward can keep alive
_pending = collections.OrderedDict()      # handle address -> _Held, for forwards whose backward has not run yet
_lock = threading.Lock()


class _Held:
    """What one forward call leaves for its backward."""
    __slots__ = ("state", "inputs", "transition", "input_lengths")

    def __init__(self, state, inputs, transition, input_lengths=None):
        self.state, self.inputs, self.transition, self.input_lengths = state, inputs, transition, input_lengths


def _handle(like, *shape):
    """A tensor that has `shape`, the dtype / device of `like` and four bytes of memory."""
    return torch.empty(1, dtype=like.dtype, device=like.device).expand(*shape)


def _hold(handle, held):
    handle._asg_held = held
    with _lock:
        _pending[handle.data_ptr()] = held
        while len(_pending) > _MAX_PENDING:
            _pending.popitem(last=False)


def _take(handle):
    held = getattr(handle, "_asg_held", None)
    with _lock:
        reg = _pending.pop(handle.data_ptr(), None)
    held = held if held is not None else reg
    if held is None:
        raise RuntimeError("torch_asg_native (HIP shim): this tensor does not come from one of this module's forward "
                           "functions (saved-tensor hooks that replace tensors are not supported on this route)")
    return held


# ---- serial route (asg.py:7-55) -----------------------------------------------------------------------------------
def fully_connected_forward(inputs, transition, input_lengths, batch_input_len, num_batches, num_labels):
    """-> (scores [B], alpha, beta, path_contrib)   (fully_connected_lattice.h:41-48)"""
    scores, state = native().full_forward(inputs, transition, input_lengths)
    alpha = _handle(inputs, batch_input_len, num_batches, num_labels)
    _hold(alpha, _Held(state, inputs, transition, input_lengths))
    return scores, alpha, _handle(inputs, batch_input_len, num_batches, num_labels), inputs.new_empty(0)


def fully_connected_backward(grad_out, alpha, beta, path_contrib, batch_input_len, num_batches, num_labels):
    """-> (grad_transition [N,N], grad_inputs [T,B,N])   (fully_connected_lattice.h:51-60)"""
    h = _take(alpha)
    return native().full_backward(h.state, grad_out, h.inputs, h.transition, h.input_lengths)


def force_aligned_forward(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                          num_labels, batch_output_len):
    """-> (scores [B], alpha, beta, path_contrib)   (force_aligned_lattice.h:42-53)"""
    scores, state = native().aligned_forward(inputs, outputs, transition, input_lengths, output_lengths)
    alpha = _handle(inputs, batch_input_len, num_batches, batch_output_len)
    _hold(alpha, _Held(state, inputs, transition))
    return scores, alpha, _handle(inputs, batch_input_len, num_batches, batch_output_len), inputs.new_empty(0)


def force_aligned_backward(grad_out, alpha, beta, path_contrib, outputs, input_lengths, output_lengths, batch_input_len,
                           num_batches, num_labels, batch_output_len):
    """-> (grad_transition, grad_inputs)   (force_aligned_lattice.h:56-69)"""
    h = _take(alpha)
    return native().aligned_backward(h.state, grad_out, h.inputs, outputs, h.transition, input_lengths, output_lengths)


# ---- GPU fast route (asg.py:58-97; streamlined_fast_gpu.h:17-68) ------------------------------------------------------
def fast_asg_gpu_forward_only(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                              num_labels, batch_output_len):
    """-> full - aligned [B]"""
    full, aligned = native().forward_only(inputs, outputs, transition, input_lengths, output_lengths, _lib.FLAG_SINGLE_LAUNCH)
    return full - aligned


def fast_asg_gpu_forward(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                         num_labels, batch_output_len):
    """-> (full_scores, aligned_scores, full_gamma, aligned_gamma, full_path_contrib, aligned_path_contrib)"""
    full, aligned, state = native().forward(inputs, outputs, transition, input_lengths, output_lengths, _lib.FLAG_SINGLE_LAUNCH)
    gamma = _handle(inputs, batch_input_len, num_batches, num_labels)
    _hold(gamma, _Held(state, inputs, transition))
    e = inputs.new_empty(0)
    return full, aligned, gamma, _handle(inputs, batch_input_len, num_batches, batch_output_len), e, e


def fast_asg_gpu_backward(grad_out_full, grad_out_aligned, full_gamma, aligned_gamma, full_path_contrib,
                          aligned_path_contrib, outputs, input_lengths, output_lengths, batch_input_len, num_batches,
                          num_labels, batch_output_len):
    """-> (grad_transition, grad_inputs)"""
    h = _take(full_gamma)
    return native().backward(h.state, grad_out_full, grad_out_aligned, h.inputs, outputs, h.transition, input_lengths,
                             output_lengths)


# ---- installation -------------------------------------------------------------------------------------------------
def install():
    """Make `import torch_asg_native` resolve to this module (before the reference's `torch_asg` is imported)."""
    sys.modules["torch_asg_native"] = sys.modules[__name__]
    return sys.modules[__name__]


def uninstall():
    if sys.modules.get("torch_asg_native") is sys.modules[__name__]:
        del sys.modules["torch_asg_native"]
