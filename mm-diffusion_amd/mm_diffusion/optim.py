"""Flat-buffer AdamW + EMA for the training step (reference: th.optim.AdamW in TrainLoop, multimodal_train_util.py:151-160,
EMA update nn.py:128-138).  All parameters live as views of ONE fp32 buffer, so the optimizer step is one kernel launch
(mmd_adamw_step) and the data-parallel gradient all-reduce is one RCCL call on one flat buffer (the reference's DDP
issues 5 buckets; its sync_params 1046 broadcasts)."""
import ctypes

import torch
import torch.distributed as dist

from . import _hip as H
from . import ops


class _PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("bwd", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("nt", ctypes.c_int), ("block_start", ctypes.c_int)]


def _pack_blocks(Cout, Cin, nt):
    """Workgroups of one weight in mmd_pack_conv_weights / mmd_unpack_conv_grads (tiles of 32 x (216 / taps) channels x all taps)."""
    n = H.lib().mmd_pack_blocks(int(Cout), int(Cin), int(nt))
    if n <= 0:
        raise H.MMDError(f"conv weight [{Cout}, {Cin}, {nt} taps]: more taps than the pack kernels stage")
    return n


class WeightPacker:
    """Keeps the GEMM-operand copies of every conv weight (forward [Cout, tap*Cin] and data-gradient [Cin, tap*Cout], in the
    activation dtype) current with ONE kernel launch per optimizer step (mmd_pack_conv_weights) and publishes them as
    `param._mmd_packed` for train_ops.ConvFn.  Parameters must stay where they are (views of FlatAdamW's flat buffer)."""

    def __init__(self, params, dtype, bucket_of=None):
        """bucket_of: parameter index -> gradient bucket (FlatAdamW), so the packed accumulators can be folded bucket by bucket."""
        self.dtype = dtype
        convs = [p for p in params if p.dim() >= 3]
        conv_bucket = [0 if bucket_of is None else bucket_of[i] for i, p in enumerate(params) if p.dim() >= 3]
        self.n = len(convs)
        if not self.n:
            return
        dev = convs[0].device
        total = sum(p.numel() for p in convs)
        self.fwd = torch.empty(total, dtype=dtype, device=dev)
        self.bwd = torch.empty(total, dtype=dtype, device=dev)
        descs = (_PackDesc * self.n)()
        off = blocks = 0
        es = self.fwd.element_size()
        for i, p in enumerate(convs):
            Cout, Cin = p.shape[0], p.shape[1]
            nt = p.numel() // (Cout * Cin)
            descs[i] = _PackDesc(p.data_ptr(), self.fwd.data_ptr() + off * es, self.bwd.data_ptr() + off * es, Cout, Cin, nt, blocks)
            p._mmd_packed = (self.fwd[off:off + p.numel()].view(Cout, nt * Cin), self.bwd[off:off + p.numel()].view(Cin, nt * Cout))
            off += p.numel()
            blocks += _pack_blocks(Cout, Cin, nt)
        self.blocks = blocks
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        self.descs = raw.to(dev)
        self.refresh()
        # gradient side: wgrad accumulates in the packed layout (coalesced atomics) and one launch per step folds it into .grad
        self.gpacked = torch.zeros(total, dtype=torch.float32, device=dev)
        gdescs = (_PackDesc * self.n)()
        off = blocks = 0
        for i, p in enumerate(convs):
            Cout, Cin = p.shape[0], p.shape[1]
            nt = p.numel() // (Cout * Cin)
            gdescs[i] = _PackDesc(p.grad.data_ptr(), self.gpacked.data_ptr() + off * 4, None, Cout, Cin, nt, blocks)
            p._mmd_wgrad = self.gpacked[off:off + p.numel()].view(Cout, nt * Cin)
            off += p.numel()
            blocks += _pack_blocks(Cout, Cin, nt)
        self.gdescs = torch.frombuffer(bytearray(bytes(gdescs)), dtype=torch.uint8).to(dev)
        # the same descriptors grouped by gradient bucket (block_start relative to the bucket's first block)
        self.bucket_descs = {}
        for bk in sorted(set(conv_bucket)):
            idx = [i for i, b in enumerate(conv_bucket) if b == bk]
            sub = (_PackDesc * len(idx))()
            blocks = 0
            for j, i in enumerate(idx):
                d = gdescs[i]
                sub[j] = _PackDesc(d.src, d.fwd, d.bwd, d.Cout, d.Cin, d.nt, blocks)
                blocks += _pack_blocks(d.Cout, d.Cin, d.nt)
            self.bucket_descs[bk] = (torch.frombuffer(bytearray(bytes(sub)), dtype=torch.uint8).to(dev), len(idx), blocks)

    def fold_grads(self, bucket=None):
        """.grad += packed wgrad accumulators (then cleared); call once after backward, before the all-reduce / optimizer step.
        bucket: only the conv weights of that gradient bucket."""
        if not self.n:
            return
        if bucket is None:
            H.call("mmd_unpack_conv_grads", self.gdescs.data_ptr(), self.n, self.blocks, H.stream_handle())
        elif bucket in self.bucket_descs:
            d, n, blocks = self.bucket_descs[bucket]
            H.call("mmd_unpack_conv_grads", d.data_ptr(), n, blocks, H.stream_handle())

    def refresh(self):
        if self.n:
            H.call("mmd_pack_conv_weights", H.BF16 if self.dtype == torch.bfloat16 else H.F32, self.descs.data_ptr(), self.n, self.blocks,
                   H.stream_handle())


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_rates=(), pack_dtype=None, grad_buckets=4,
                 grad_payload="fp32"):
        """pack_dtype: the model's activation dtype (model.dtype) - enables the one-launch conv-weight re-pack after every step.
        grad_buckets: the flat gradient is reduced over the ranks in this many contiguous buckets (parameter order = forward order,
        so backward completes the LAST bucket first); with overlap armed (arm_overlap) a bucket's all-reduce is issued on a side
        stream as soon as the backward has passed its last parameter, under the rest of the backward (the reference: DDP with
        128 MB buckets, multimodal_train_util.py:127-136).  grad_payload "bf16": buckets travel as bf16 (half the xGMI bytes)."""
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:           # re-home every parameter (and its .grad) into the flat buffers
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.ema_rates = list(ema_rates)
        self.ema_params = [self.flat.clone() for _ in self.ema_rates]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.steps = 0
        # ---- gradient buckets: contiguous ranges of the flat buffer cut at parameter boundaries, ~equal sizes
        sizes = [p.numel() for p in self.params]
        nb = max(1, min(int(grad_buckets), len(self.params)))
        self.bucket_of, self.buckets = [], []          # parameter index -> bucket; bucket -> [elem lo, elem hi, params]
        tgt, acc, start = n / nb, 0, 0
        for i, k in enumerate(sizes):
            b = len(self.buckets)
            self.bucket_of.append(b)
            acc += k
            if (acc >= tgt * (b + 1) and b < nb - 1) or i == len(sizes) - 1:
                self.buckets.append([start, acc, i + 1 - sum(bk[2] for bk in self.buckets)])
                start = acc
        if grad_payload not in ("fp32", "bf16"):
            raise ValueError("grad_payload must be 'fp32' or 'bf16'")
        self.grad_payload = grad_payload
        self._armed, self._left, self._ready, self._inflight, self._side = False, [], [], [], None
        for i, p in enumerate(self.params):
            p._mmd_opt = (self, i)                     # train_ops._grad_slot reports "this parameter's gradient kernel is enqueued"
        self.packer = WeightPacker(self.params, pack_dtype, self.bucket_of) if pack_dtype is not None and self.params[0].is_cuda else None

    def zero_grad(self):
        self.grad.zero_()

    def fold_grads(self):
        """Fold the packed conv-weight gradient accumulators into .grad (once per step: all_reduce_grads() or step() does it)."""
        if self.packer is not None and not getattr(self, "_folded", False):
            done = set(getattr(self, "_folded_buckets", ()))
            if not done:
                self.packer.fold_grads()
            else:
                for b in range(len(self.buckets)):
                    if b not in done:
                        self.packer.fold_grads(b)
        self._folded, self._folded_buckets = True, set()

    # ------------------------------------------------------------------ data-parallel gradient reduction
    def arm_overlap(self):
        """Call before the LAST backward of a step (after the last microbatch's forward): from now on every parameter whose
        gradient kernel has been enqueued is counted, and a bucket whose parameters are all in is reduced at once on a side stream
        while the backward continues.  all_reduce_grads() then only reduces what is left and waits."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        self._armed = True
        self._left = [bk[2] for bk in self.buckets]
        self._ready, self._inflight, self._seen = [], [], set()
        self._folded_buckets = set()

    def _flush_ready(self):
        """Launch the buckets that became complete at EARLIER _grad_slot calls (their gradient kernels are on the stream by now).
        Called ONCE per backward kernel wrapper, before any of that kernel's parameters is marked: a kernel that completes the
        gradients of two parameters (weight + bias, gamma + beta) must not see the bucket its FIRST parameter completes launched while
        its own launch is still to come (the bucket would be folded and reduced without this step's contribution)."""
        if not self._armed:
            return
        while self._ready:
            self._launch_bucket(self._ready.pop(0))

    def _param_done(self, i):
        """train_ops._grad_slot: the backward kernel that completes parameter i's gradient is about to be enqueued.  Only marks: a
        bucket completed here is launched by the next _flush_ready() (the next kernel wrapper) or by all_reduce_grads()."""
        if not self._armed:
            return
        if i in self._seen:
            return
        self._seen.add(i)
        b = self.bucket_of[i]
        self._left[b] -= 1
        if self._left[b] == 0:
            self._ready.append(b)

    def _launch_bucket(self, b):
        lo, hi, _ = self.buckets[b]
        world = dist.get_world_size()
        on_gpu = self.grad.is_cuda
        if self.packer is not None:
            self.packer.fold_grads(b)                  # this bucket's packed conv-weight accumulators -> .grad
            self._folded_buckets.add(b)
        view = self.grad[lo:hi]
        if on_gpu:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.grad.device)
            self._side.wait_stream(torch.cuda.current_stream(self.grad.device))
            ctx = torch.cuda.stream(self._side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.grad_payload == "bf16" and on_gpu:
                pay = self._payload(b)
                ops.cast(view, pay)                    # fp32 -> bf16 (libmmd), reduced as bf16, widened back after the wait
                work = dist.all_reduce(pay, op=dist.ReduceOp.SUM, async_op=True)
            else:
                pay = None
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)
        self._inflight.append((b, work, pay, world))

    def _payload(self, b):
        if not hasattr(self, "_pay"):
            self._pay = {}
        if b not in self._pay:
            lo, hi, _ = self.buckets[b]
            self._pay[b] = torch.empty(hi - lo, dtype=torch.bfloat16, device=self.grad.device)
        return self._pay[b]

    def all_reduce_grads(self):
        """Data-parallel gradient mean over ranks (RCCL over xGMI): the buckets not yet reduced under the backward (all of them
        when overlap was not armed) are reduced now, last bucket first; then everything in flight is awaited and scaled by
        1 / world.  Folds the packed conv-weight gradient accumulators into .grad first."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            self.fold_grads()
            self._armed = False
            return
        if not self._armed:
            self._inflight, self._folded_buckets = [], set()
        started = {b for b, *_ in self._inflight}
        self._ready = []
        for b in reversed(range(len(self.buckets))):
            if b not in started:
                self._launch_bucket(b)
        self._armed = False
        world = dist.get_world_size()
        for b, work, pay, _ in self._inflight:
            work.wait()                                # the current stream waits for the collective
            lo, hi, _n = self.buckets[b]
            if pay is not None:
                ops.cast(pay, self.grad[lo:hi], scale=1.0 / world)
            else:
                self.grad[lo:hi].div_(world)
        if self._side is not None:
            torch.cuda.current_stream(self.grad.device).wait_stream(self._side)
        self._inflight = []
        self._folded, self._folded_buckets = True, set()

    def step(self):
        self.fold_grads()             # no-op when all_reduce_grads already folded (the accumulators are cleared by the fold)
        self.steps += 1
        # autograd accumulates into p.grad in place, so self.grad already holds the flat gradient
        lo, hi = self.grad.data_ptr(), self.grad.data_ptr() + self.grad.numel() * 4
        for p in self.params:
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError("a parameter's .grad was replaced or dropped; FlatAdamW needs in-place gradient accumulation")
        ema0 = self.ema_params[0] if self.ema_params else None
        ops.adamw_step(self.flat, self.grad, self.m, self.v, ema0, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.weight_decay, self.steps, ema_rate=self.ema_rates[0] if self.ema_rates else 0.0)
        for rate, ema in zip(self.ema_rates[1:], self.ema_params[1:]):
            ema.mul_(rate).add_(self.flat, alpha=1 - rate)
        if self.packer is not None:
            self.packer.refresh()
        self._folded = False
        # the kernel wrote the parameters through the flat buffer: tell autograd / the inference engine (which keys its packed
        # weights on the parameters' version counters, engine.UNetEngine.stale) that every parameter changed
        bump = getattr(torch._C, "_increment_version", None)
        if bump is not None:
            bump(self.params)      # ONE call on the list: handed a single tensor it iterates over it (unbind per row: 0.27 s per step)
