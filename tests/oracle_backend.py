"""TEST-ONLY stand-in for torch_asg_amd.asg.HipBackend built on the CPU oracle.

Lets the CPU test-suite exercise the host logic of torch_asg_amd (length defaults, truncation, routing,
reductions, autograd plumbing, batch sharding + all-reduce) without a GPU.  It is injected by the tests
with `torch_asg_amd.asg._backend = OracleBackend()`; the shipped package never references it.
"""
import numpy as np
import torch

from oracle import asg_oracle as orc


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


class OracleBackend:
    def full_forward(self, inputs, transition, input_lengths, flags=0):
        s, a, b = orc.full_forward(_np(inputs), _np(transition), _np(input_lengths))
        state = torch.from_numpy(np.stack([a, b]))
        return torch.from_numpy(s), state

    def full_backward(self, state, grad_out, inputs, transition, input_lengths):
        a, b = state[0].numpy(), state[1].numpy()
        gtr, gin = orc.full_backward(_np(grad_out), a, b, _np(inputs), _np(transition))
        return torch.from_numpy(gtr), torch.from_numpy(gin)

    def aligned_forward(self, inputs, targets, transition, input_lengths, target_lengths, flags=0):
        s, a, b = orc.aligned_forward(_np(inputs), _np(targets), _np(transition), _np(input_lengths), _np(target_lengths))
        return torch.from_numpy(s), torch.from_numpy(np.stack([a, b]))

    def aligned_backward(self, state, grad_out, inputs, targets, transition, input_lengths, target_lengths):
        a, b = state[0].numpy(), state[1].numpy()
        gtr, gin = orc.aligned_backward(_np(grad_out), a, b, _np(targets), _np(transition), _np(input_lengths),
                                        _np(target_lengths), inputs.shape[2])
        return torch.from_numpy(gtr), torch.from_numpy(gin)

    def forward(self, inputs, targets, transition, input_lengths, target_lengths, flags=0):
        fs, fstate = self.full_forward(inputs, transition, input_lengths)
        as_, astate = self.aligned_forward(inputs, targets, transition, input_lengths, target_lengths)
        T, B, N = inputs.shape
        S = targets.shape[1]
        state = torch.cat([fstate.reshape(-1), astate.reshape(-1)])
        return fs, as_, state

    def forward_only(self, inputs, targets, transition, input_lengths, target_lengths, flags=0):
        fs, _ = self.full_forward(inputs, transition, input_lengths)
        as_, _ = self.aligned_forward(inputs, targets, transition, input_lengths, target_lengths)
        return fs, as_

    def backward(self, state, grad_full, grad_aligned, inputs, targets, transition, input_lengths, target_lengths,
                 flags=0):
        T, B, N = inputs.shape
        S = targets.shape[1]
        nf = 2 * T * B * N
        fstate = state[:nf].reshape(2, T, B, N)
        astate = state[nf:].reshape(2, T, B, S)
        g1, i1 = self.full_backward(fstate, grad_full, inputs, transition, input_lengths)
        g2, i2 = self.aligned_backward(astate, grad_aligned, inputs, targets, transition, input_lengths, target_lengths)
        return g1 + g2, i1 + i2

    def loss_forward(self, inputs, targets, transition, input_lengths, target_lengths, reduction, flags=0):
        from torch_asg_amd.asg import _Saved
        fs, as_, state = self.forward(inputs, targets, transition, input_lengths, target_lengths)
        per = fs - as_
        loss = per if reduction == "none" else (per.sum() if reduction == "sum" else per.mean())
        return loss, _Saved("split", (state,), None, None, None)

    def loss_backward(self, saved, tensors, grad_loss, inputs, targets, transition, input_lengths, target_lengths,
                      reduction):
        (state,) = tensors
        B = inputs.shape[1]
        g = grad_loss.reshape(-1).to(inputs.dtype)
        if reduction != "none":
            g = g.expand(B) * (1.0 / B if reduction == "mean" else 1.0)
        g = g.contiguous()
        return self.backward(state, g, -g, inputs, targets, transition, input_lengths, target_lengths)
