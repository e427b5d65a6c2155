"""The ASG training step as ONE hipGraph replay -- the mode `bench.py`'s headline is measured in, as public API.

    crit = torch_asg_amd.ASGLoss(40).cuda()
    step = torch_asg_amd.graphed(crit, (emissions, targets, input_lengths, target_lengths))
    for emissions, targets, input_lengths, target_lengths in loader:        # same shapes as the sample, any values
        loss = step(emissions, targets, input_lengths, target_lengths)      # copy in, replay: forward + backward
        upstream_output.backward(step.inputs_grad)                          # d loss / d emissions, [T,B,N]
        optimizer.step()                                                    # crit.transition.grad is set

The reference has no counterpart (its step is eager: asg.py:100-142 builds an autograd graph per call).  What is
recorded is exactly `loss = module(inputs, targets, input_lengths, target_lengths); loss.backward(grad_scale)` on static
buffers; the kernels read lengths and targets from device memory, so a replay with other values in the same buffers is
the eager step on those values, bit for bit on the deterministic routes (tests/test_hip_graphed.py).

`torch.cuda.make_graphed_callables(ASGLoss(...), sample_args)` works too (autograd-integrated, forward and backward as
two graphs); this helper records the whole step as one graph, with no autograd bookkeeping left at replay time.
"""
import torch

from . import asg as _asg


class GraphedStep:
    """A captured `module(...)` + `loss.backward(...)` over static buffers.  Build with `torch_asg_amd.graphed`.

    Attributes (all static device tensors, valid after every call):
      loss            what `module(...)` returned (scalar, or [B] with reduction='none')
      inputs          the emissions buffer [T,B,N]; an upstream network may write into it directly (then call with
                      inputs=None and nothing is copied)
      inputs_grad     d loss / d inputs (None for an evaluation-mode module)
      targets, input_lengths, target_lengths
    `module.transition.grad` is (re)assigned to the captured gradient buffer after every replay.
    """

    def __init__(self, module, sample_args, steps=1, grad_scale=None, after_step=None, warmup=3,
                 capture_error_mode="global", pool=None):
        if not isinstance(module, _asg.ASGLoss):
            raise TypeError("torch_asg_amd.graphed: an ASGLoss module is expected, got %s" % type(module).__name__)
        if steps < 1:
            raise ValueError("steps must be >= 1")
        inputs, targets = sample_args[0], sample_args[1]
        input_lengths = sample_args[2] if len(sample_args) > 2 else None
        target_lengths = sample_args[3] if len(sample_args) > 3 else None
        if not inputs.is_cuda:
            raise RuntimeError("torch_asg_amd.graphed: the sample emissions must live on a ROCm device")
        dev = inputs.device
        if module.transition.device != dev:
            raise RuntimeError("torch_asg_amd.graphed: module on %s, sample on %s" % (module.transition.device, dev))
        self.module = module
        self.steps = int(steps)
        self.backward = bool(module.training and not module.forward_only)
        T, B = inputs.shape[0], inputs.shape[1]
        # static buffers.  Missing lengths get the reference's defaults (asg.py:113-117) ONCE, here, so that no fill kernel
        # and no allocation is part of the recorded step
        self.inputs = inputs.detach().clone(memory_format=torch.contiguous_format).requires_grad_(self.backward)
        self.targets = targets.detach().to(dev).clone()
        S = self.targets.shape[1]
        self.target_lengths = (torch.full((B,), S, dtype=torch.int64, device=dev) if target_lengths is None
                               else target_lengths.detach().to(dev).clone().contiguous())
        self.input_lengths = (torch.full((B,), T, dtype=torch.int64, device=dev) if input_lengths is None
                              else input_lengths.detach().to(dev).clone().contiguous())
        gs = torch.ones(()) if grad_scale is None else (grad_scale if torch.is_tensor(grad_scale) else torch.tensor(float(grad_scale)))
        self.grad_scale = gs.detach().to(device=dev, dtype=module.transition.dtype).clone()
        self._grad_out = None                       # grad_scale in the shape of the loss, made during the warm-up: a static
                                                    # tensor, so that the recorded backward starts without a fill kernel
        self._after = after_step
        _asg.native().reserve(dev)                  # the fused step's zeroed sync pool cannot be created under capture

        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # warm-up off the capture: loads the kernels, sizes the allocator's pools
                for _ in range(max(int(warmup), 1)):
                    self._one_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            kw = {"capture_error_mode": capture_error_mode}
            if pool is not None:
                kw["pool"] = pool
            with torch.cuda.graph(self.graph, **kw):
                for _ in range(self.steps):
                    self.loss = self._one_step()
            self.loss = self.loss.detach()
            self.inputs_grad = self.inputs.grad if self.backward else None
            self._transition_grad = module.transition.grad if self.backward else None
            self.graph.replay()
            torch.cuda.synchronize(dev)

    def _one_step(self):
        m = self.module
        if not self.backward:
            with torch.no_grad():
                return m(self.inputs, self.targets, self.input_lengths, self.target_lengths)
        m.transition.grad = None
        self.inputs.grad = None
        loss = m(self.inputs, self.targets, self.input_lengths, self.target_lengths)
        if self._grad_out is None:
            self._grad_out = self.grad_scale.expand_as(loss).contiguous()
        loss.backward(self._grad_out)
        if self._after is not None:
            self._after()
        return loss

    @staticmethod
    def _fill(dst, src, what):
        if src is None or src is dst:
            return
        if tuple(src.shape) != tuple(dst.shape):
            raise RuntimeError("torch_asg_amd.graphed: %s has shape %s, the captured step was recorded for %s -- pad to the "
                               "recorded shape and say so in the lengths" % (what, tuple(src.shape), tuple(dst.shape)))
        dst.copy_(src, non_blocking=True)

    def __call__(self, inputs=None, targets=None, input_lengths=None, target_lengths=None):
        """Copy the given tensors into the static buffers (None = keep what is there), replay, return the loss."""
        with torch.no_grad():
            self._fill(self.inputs, inputs, "inputs")
            self._fill(self.targets, targets, "targets")
            self._fill(self.input_lengths, input_lengths, "input_lengths")
            self._fill(self.target_lengths, target_lengths, "target_lengths")
        self.graph.replay()
        if self.backward:
            # (an optimizer's zero_grad(set_to_none=True) drops the attribute, not the captured buffer)
            self.module.transition.grad = self._transition_grad
            self.inputs.grad = self.inputs_grad
        return self.loss

    replay = __call__


def graphed(module, sample_args, steps=1, grad_scale=None, after_step=None, warmup=3, capture_error_mode="global",
            pool=None):
    """Record `loss = module(*sample_args); loss.backward(grad_scale)` into one hipGraph over static copies of
    `sample_args` = (inputs [T,B,N], targets [B,S], input_lengths [B] | None, target_lengths [B] | None) and return
    the `GraphedStep`; calling it with tensors of the same shapes copies them in and replays.

      steps        consecutive steps recorded per replay (> 1 only makes sense for measurements: every step sees the
                   same buffers; bench.py records 10 so that a replay is long against its launch)
      grad_scale   the gradient handed to `loss.backward` (number or tensor; default 1) -- e.g. 1/world in a
                   batch-sharded run, where it enters the kernels instead of costing a launch
      after_step   called after each recorded step, inside the capture (the all-reduce of transition.grad)
    A module in evaluation mode (or forward_only=True) records the evaluation route: no gradients.
    """
    return GraphedStep(module, sample_args, steps=steps, grad_scale=grad_scale, after_step=after_step, warmup=warmup,
                       capture_error_mode=capture_error_mode, pool=pool)
