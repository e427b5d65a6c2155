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
    HIP path keeps ONE opaque state buffer and recomputes instead of reading path_contrib.  Everything the backward pass
    needs travels in the DATA of the tensors asg.py saves (so saved-tensor hooks that replace tensors -- save_on_cpu,
    checkpoint wrappers -- are fine; nothing is keyed by tensor identity or address):
      - the [T,B,N] slot whose shape asg.py reads (alpha of FCC, full_gamma of ASGGPUFast) IS the emissions tensor, and
        FAC's beta (shape never read) is; the kernels need the emissions again in the backward pass and asg.py does not
        save them on these routes;
      - path_contrib (never looked at by asg.py) is one byte buffer: [state | copy of input_lengths (FCC) | copy of transition];
      - the remaining slots are zero-stride placeholders of the documented shape (four bytes of memory).
  * CPU tensors raise (this package has no CPU implementation; the reference would run its OpenMP path).
"""
import sys

import torch

from . import _lib
from .asg import native

__all__ = ["fully_connected_forward", "fully_connected_backward", "force_aligned_forward", "force_aligned_backward",
           "fast_asg_gpu_forward_only", "fast_asg_gpu_forward", "fast_asg_gpu_backward", "install", "uninstall"]


def _handle(like, *shape):
    """A tensor that has `shape`, the dtype / device of `like` and four bytes of memory."""
    return torch.empty(1, dtype=like.dtype, device=like.device).expand(*shape)


_MAGIC = 0x4153475F53544154          # "ASG_STAT": first word of the tail behind a state buffer this module allocated
STRICT = False                       # True: _unpack also checks the magic word (a device -> host read: one synchronisation per backward)


def _tail_bytes(like, num_batches, num_labels, with_lengths):
    """Bytes behind the state: [magic, state bytes] (2 x int64) | lengths (int64 x B, optional) | transition (N x N of like.dtype), padded
    to 8 bytes."""
    n = 16 + (num_batches * 8 if with_lengths else 0) + num_labels * num_labels * like.element_size()
    return (n + 7) // 8 * 8


def _fill_tail(state, tail, like, transition, lengths=None):
    """Write the tail of `state` (allocated with `tail` extra bytes by HipBackend.*forward(tail_bytes=...)) IN PLACE: the lattice state --
    14.5 MB at T=400 B=64 N=40 -- is not copied, only these few kilobytes are.  Returns `state`: what asg.py saves as path_contrib."""
    nstate = state.numel() - tail
    t = state[nstate:]
    # (written on the device: a copy from a pageable host tensor would stall the host behind everything queued on the stream, and
    # is illegal while the stream is being captured)
    head = t[:16].view(torch.int64)
    head[0].fill_(_MAGIC)
    head[1].fill_(nstate)
    off = 16
    if lengths is not None:
        nl = lengths.numel() * 8
        t[off:off + nl].view(torch.int64).copy_(lengths.to(dtype=torch.int64).reshape(-1), non_blocking=True)
        off += nl
    nt = transition.numel() * like.element_size()
    t[off:off + nt].view(like.dtype).copy_(transition.detach().to(like.dtype).reshape(-1), non_blocking=True)
    return state


def _unpack(packed, like, num_batches, num_labels, with_lengths):
    """-> (state, transition [N,N], lengths [B] or None) as views of `packed`; `like` gives the dtype."""
    tail = _tail_bytes(like, num_batches, num_labels, with_lengths)
    rest = (packed.numel() - tail) if (packed.dtype == torch.uint8 and packed.dim() == 1) else -1
    if rest <= 0 or rest % 256 != 0:       # (the state buffer is a whole number of 256-byte units)
        raise RuntimeError("torch_asg_native (HIP shim): path_contrib does not come from this module's forward functions")
    t = packed[rest:]
    if STRICT:
        head = t[:16].view(torch.int64).cpu()
        if int(head[0]) != _MAGIC or int(head[1]) != rest:
            raise RuntimeError("torch_asg_native (HIP shim): path_contrib does not come from this module's forward functions")
    off = 16
    lengths = None
    if with_lengths:
        lengths = t[off:off + num_batches * 8].view(torch.int64)
        off += num_batches * 8
    nt = num_labels * num_labels * like.element_size()
    transition = t[off:off + nt].view(like.dtype).view(num_labels, num_labels)
    return packed[:rest], transition, lengths


# ---- serial route (asg.py:7-55) -----------------------------------------------------------------------------------
def fully_connected_forward(inputs, transition, input_lengths, batch_input_len, num_batches, num_labels):
    """-> (scores [B], alpha, beta, path_contrib)   (fully_connected_lattice.h:41-48)"""
    tail = _tail_bytes(inputs, num_batches, num_labels, True)
    scores, state = native().full_forward(inputs, transition, input_lengths, tail_bytes=tail)
    if input_lengths is None:
        input_lengths = torch.full((num_batches,), batch_input_len, dtype=torch.int64, device=inputs.device)
    return scores, inputs, _handle(inputs, batch_input_len, num_batches, num_labels), _fill_tail(state, tail, inputs, transition, input_lengths)


def fully_connected_backward(grad_out, alpha, beta, path_contrib, batch_input_len, num_batches, num_labels):
    """-> (grad_transition [N,N], grad_inputs [T,B,N])   (fully_connected_lattice.h:51-60)"""
    state, transition, lengths = _unpack(path_contrib, alpha, num_batches, num_labels, True)
    return native().full_backward(state, grad_out, alpha, transition, lengths)


def force_aligned_forward(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                          num_labels, batch_output_len):
    """-> (scores [B], alpha, beta, path_contrib)   (force_aligned_lattice.h:42-53)"""
    tail = _tail_bytes(inputs, num_batches, num_labels, False)
    scores, state = native().aligned_forward(inputs, outputs, transition, input_lengths, output_lengths, tail_bytes=tail)
    return scores, _handle(inputs, batch_input_len, num_batches, batch_output_len), inputs, _fill_tail(state, tail, inputs, transition)


def force_aligned_backward(grad_out, alpha, beta, path_contrib, outputs, input_lengths, output_lengths, batch_input_len,
                           num_batches, num_labels, batch_output_len):
    """-> (grad_transition, grad_inputs)   (force_aligned_lattice.h:56-69)"""
    state, transition, _ = _unpack(path_contrib, beta, num_batches, num_labels, False)
    return native().aligned_backward(state, grad_out, beta, outputs, transition, input_lengths, output_lengths)


# ---- GPU fast route (asg.py:58-97; streamlined_fast_gpu.h:17-68) ------------------------------------------------------
def fast_asg_gpu_forward_only(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                              num_labels, batch_output_len):
    """-> full - aligned [B]"""
    full, aligned = native().forward_only(inputs, outputs, transition, input_lengths, output_lengths, _lib.FLAG_SINGLE_LAUNCH)
    return full - aligned


def fast_asg_gpu_forward(inputs, outputs, transition, input_lengths, output_lengths, batch_input_len, num_batches,
                         num_labels, batch_output_len):
    """-> (full_scores, aligned_scores, full_gamma, aligned_gamma, full_path_contrib, aligned_path_contrib)"""
    tail = _tail_bytes(inputs, num_batches, num_labels, False)
    full, aligned, state = native().forward(inputs, outputs, transition, input_lengths, output_lengths, _lib.FLAG_SINGLE_LAUNCH, tail_bytes=tail)
    return (full, aligned, inputs, _handle(inputs, batch_input_len, num_batches, batch_output_len),
            _fill_tail(state, tail, inputs, transition), inputs.new_empty(0))


def fast_asg_gpu_backward(grad_out_full, grad_out_aligned, full_gamma, aligned_gamma, full_path_contrib,
                          aligned_path_contrib, outputs, input_lengths, output_lengths, batch_input_len, num_batches,
                          num_labels, batch_output_len):
    """-> (grad_transition, grad_inputs)"""
    state, transition, _ = _unpack(full_path_contrib, full_gamma, num_batches, num_labels, False)
    return native().backward(state, grad_out_full, grad_out_aligned, full_gamma, outputs, transition, input_lengths,
                             output_lengths)


# ---- installation -------------------------------------------------------------------------------------------------
def install():
    """Make `import torch_asg_native` resolve to this module (before the reference's `torch_asg` is imported)."""
    sys.modules["torch_asg_native"] = sys.modules[__name__]
    return sys.modules[__name__]


def uninstall():
    if sys.modules.get("torch_asg_native") is sys.modules[__name__]:
        del sys.modules["torch_asg_native"]
