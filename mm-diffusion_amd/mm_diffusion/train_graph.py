"""Whole-step graph capture of the training step: zero_grad + q_sample + U-Net forward + loss + backward (incl. the
recompute of the checkpointed cross-attention blocks) are recorded ONCE with `torch.cuda.graph` and replayed per step.

Why: at per-GPU batch 8 the eager step is host-bound - ~1500 autograd Function applications and as many backward calls cost
~160 ms of Python per step while the kernels sum to ~125 ms.  A replay costs one launch.  What changes per step lives in
static device buffers refreshed before the replay: the batch, the timesteps, their importance weights, the q_sample noise and
the cross-attention window shifts (the kernels already read the shift from a device int; the forward draws and the
recompute's re-draws - reference nn.py:262-270 - are separate slots filled in execution order).  Dropout masks come from
torch's graph-safe Philox stream.  The optimizer step (bias-correction scalars change every step), the gradient all-reduce and
the one-launch weight re-pack stay outside the graph.
"""
import random

import torch as th

from . import _hip as H


class ShiftSlots:
    """Stands in for `random.randint` as `model.shift_source`: hands out 1-element views of a persistent device table in call
    order (train_ops.CrossAttnFn takes the view as its shift_dev) and remembers each slot's (lo, hi)."""

    def __init__(self, device, capacity=256):
        self.table = th.zeros(capacity, dtype=th.int32, device=device)
        self._up = H.Staged(self.table)        # pinned ring: the host runs ahead of the replayed steps
        self.ranges = []
        self.idx = 0
        self.frozen = False              # True while capturing / replaying: the slot sequence must repeat exactly

    def begin_step(self):
        self.idx = 0

    def __call__(self, lo, hi):
        i = self.idx
        self.idx += 1
        if self.frozen:
            if i >= len(self.ranges) or self.ranges[i] != (lo, hi):
                raise RuntimeError("the cross-attention shift sequence changed between the recorded and the captured step")
        else:
            if i == len(self.ranges):
                self.ranges.append((lo, hi))
            v = random.randint(lo, hi)
            self.table[i:i + 1].fill_(v)                  # eager warm-up steps: value written in stream order
        return self.table[i:i + 1]

    def randomize(self):
        """Fresh draws for every slot (same `random.randint(lo, hi)` calls, same order as an eager step would make)."""
        host = self._up.host()
        for i, (lo, hi) in enumerate(self.ranges):
            host[i] = random.randint(lo, hi)
        self._up.push()


class GraphedTrainStep:
    def __init__(self, model, diffusion, opt, batch, warmup=2):
        """batch: {"video": [B,F,C,H,W], "audio": [B,C,L]} example tensors (shapes fix the graph)."""
        self.model, self.diffusion, self.opt = model, diffusion, opt
        dev = next(model.parameters()).device
        self.x0 = {k: th.zeros(v.shape, dtype=th.float32, device=dev) for k, v in batch.items()}
        self.noise = {k: th.zeros_like(v) for k, v in self.x0.items()}
        B = self.x0["video"].shape[0]
        self.t = th.zeros(B, dtype=th.int64, device=dev)
        self.weights = th.ones(B, dtype=th.float32, device=dev)
        self.slots = ShiftSlots(dev)
        self._prev_source = getattr(model, "shift_source", None)
        model.shift_source = self.slots
        self.losses = None
        self.graph = None
        self._warmup = warmup

    def _forward_backward(self):
        self.slots.begin_step()
        self.opt.zero_grad()
        losses = self.diffusion.multimodal_training_losses(self.model, self.x0, self.t, noise=self.noise)
        (losses["loss"] * self.weights).mean().backward()
        return {k: v.detach() for k, v in losses.items()}

    def _capture(self):
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):
            for _ in range(self._warmup):                   # allocator / autotune / lazy-init warm-up, records the slot ranges
                self._forward_backward()
        th.cuda.current_stream().wait_stream(side)
        th.cuda.synchronize()
        self.slots.frozen = True
        self.slots.randomize()
        self.graph = th.cuda.CUDAGraph()
        with th.cuda.graph(self.graph):
            self.losses = self._forward_backward()

    def step(self, batch, t, weights=None, noise=None):
        """One training step on `batch` at timesteps `t`; returns the per-sample loss terms (device tensors, valid until the
        next step)."""
        for k in self.x0:
            self.x0[k].copy_(batch[k], non_blocking=True)
            if noise is None:
                self.noise[k].normal_()
            else:
                self.noise[k].copy_(noise[k], non_blocking=True)
        self.t.copy_(t, non_blocking=True)
        if weights is None:
            self.weights.fill_(1.0)
        else:
            self.weights.copy_(weights, non_blocking=True)
        if self.graph is None:
            self._capture()
        self.slots.randomize()
        self.graph.replay()
        self.opt.all_reduce_grads()
        self.opt.step()
        return self.losses

    def close(self):
        self.model.shift_source = self._prev_source
