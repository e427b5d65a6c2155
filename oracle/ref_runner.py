"""oracle/ref_runner.py -- drive the REAL reference's compiled CPU path. TEST INFRASTRUCTURE ONLY.

Loads oracle/_ref/torch_asg_native.so (built by oracle/build_ref.py from the reference's own
C++ sources) and calls its four CPU entry points (/root/reference/torch_asg/native/
extension.cpp:16-19) in the order the reference's "serial" route does
(/root/reference/torch_asg/asg.py:124-128 + the FCC/FAC backward at asg.py:26-34,48-55).
The reference's Python file itself never travels to the GPU box; this harness is our own.

Used for (a) validating oracle/asg_oracle.c live, (b) bench.py's cpu_baseline
(kind = "reference").  Never imported by torch_asg_amd/.
"""
import importlib.util
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "torch_asg_native.so")
_MOD = None


def available():
    return os.path.exists(_SO)


def native():
    global _MOD
    if _MOD is None:
        if not available():
            raise RuntimeError("oracle/_ref/torch_asg_native.so missing: run `python oracle/build_ref.py` "
                               "in the container that has /root/reference")
        spec = importlib.util.spec_from_file_location("torch_asg_native", _SO)
        _MOD = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_MOD)
    return _MOD


def _prep(inputs, targets, input_lengths, target_lengths):
    T, B, N = inputs.shape
    S = targets.shape[1]
    if target_lengths is None:
        target_lengths = torch.full((B,), S, dtype=torch.int64)
    if input_lengths is None:
        input_lengths = torch.full((B,), T, dtype=torch.int64)
    if S > T:                       # asg.py:119-122
        S = T
        targets = targets[:, :S]
        target_lengths = torch.clamp(target_lengths, max=S)
    return targets, input_lengths, target_lengths, (T, B, N, S)


def forward(inputs, targets, transition, input_lengths=None, target_lengths=None):
    """Reference CPU forward. Returns (full_scores, aligned_scores, saved) with saved for backward()."""
    m = native()
    targets, il, tl, (T, B, N, S) = _prep(inputs, targets, input_lengths, target_lengths)
    a_scores, a_alpha, a_beta, a_pc = m.force_aligned_forward(inputs, targets, transition, il, tl, T, B, N, S)
    f_scores, f_alpha, f_beta, f_pc = m.fully_connected_forward(inputs, transition, il, T, B, N)
    saved = dict(full=(f_alpha, f_beta, f_pc), aligned=(a_alpha, a_beta, a_pc),
                 targets=targets, il=il, tl=tl, dims=(T, B, N, S))
    return f_scores, a_scores, saved


def backward(grad_per_utt, saved):
    """Reference CPU backward for loss[b] = full[b] - aligned[b] with d(total)/d(loss[b]) = grad_per_utt[b]."""
    m = native()
    T, B, N, S = saved["dims"]
    f_alpha, f_beta, f_pc = saved["full"]
    a_alpha, a_beta, a_pc = saved["aligned"]
    g = grad_per_utt.contiguous()
    gtr_f, gin_f = m.fully_connected_backward(g, f_alpha, f_beta, f_pc, T, B, N)
    ng = (-g).contiguous()
    gtr_a, gin_a = m.force_aligned_backward(ng, a_alpha, a_beta, a_pc, saved["targets"], saved["il"],
                                            saved["tl"], T, B, N, S)
    return gtr_f + gtr_a, gin_f + gin_a


def asg_loss(inputs, targets, transition, input_lengths=None, target_lengths=None, reduction="mean"):
    """Full reference fwd+bwd (grad of the reduced loss = 1). Returns dict of torch tensors."""
    full, ali, saved = forward(inputs, targets, transition, input_lengths, target_lengths)
    per_utt = full - ali
    B = per_utt.shape[0]
    if reduction == "mean":
        loss, g = per_utt.mean(), torch.full((B,), 1.0 / B, dtype=inputs.dtype)
    elif reduction == "sum":
        loss, g = per_utt.sum(), torch.ones(B, dtype=inputs.dtype)
    else:
        loss, g = per_utt, torch.ones(B, dtype=inputs.dtype)
    gtr, gin = backward(g, saved)
    return dict(loss=loss, loss_per_utt=per_utt, full_scores=full, aligned_scores=ali,
                grad_inputs=gin, grad_transition=gtr)
