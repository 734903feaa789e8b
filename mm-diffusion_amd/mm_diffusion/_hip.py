"""ctypes binding of libmmd.so (include/mmd.h).  There is NO fallback: if the library is missing or a
call fails this raises - the product path never silently computes on the CPU or with stock torch ops."""
import ctypes as C
import gc
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMD_LIB: an ablation / experiment build of the library (tools/ only; the product never sets it)
LIB_PATH = os.environ.get("MMD_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmmd.so")

F32, BF16 = 0, 1
_lib = None

i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p


_PROTOS = {
    "mmd_version": (C.c_int, []),
    "mmd_last_error": (C.c_char_p, []),
    "mmd_debug_install_crash_handler": (i32, []),
    "mmd_graph_begin": (i32, [vp]),
    "mmd_graph_end": (i32, [vp, C.POINTER(vp)]),
    "mmd_graph_launch": (i32, [vp, vp]),
    "mmd_graph_destroy": (i32, [vp]),
    "mmd_stream_create": (i32, [C.POINTER(vp)]),
    "mmd_stream_sync": (i32, [vp]),
    "mmd_stream_destroy": (i32, [vp]),
    "mmd_event_create": (i32, [C.POINTER(vp)]),
    "mmd_event_record": (i32, [vp, vp]),
    "mmd_stream_wait_event": (i32, [vp, vp]),
    "mmd_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "mmd_event_destroy": (i32, [vp]),
    "mmd_temb_fwd": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "mmd_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "mmd_gn_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "mmd_gn_stats": (i32, [i32, vp, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, vp, i64, f32, vp, vp, vp, vp, vp]),
    "mmd_gn_small": (i32, [i32, vp, i64, vp, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, f32, i32, vp]),
    "mmd_gn_group": (i32, [i32, vp, i64, vp, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, vp, i64, f32, i32, vp, vp, vp, vp]),
    "mmd_gn_apply": (i32, [i32, vp, i64, vp, i64, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, i32, vp]),
    "mmd_add_rowbias": (i32, [i32, vp, i64, i64, i32, i64, vp, i64, vp]),
    "mmd_colsum_slices": (i32, [i32, vp, i64, i32, i64, i32, vp, i64, vp]),
    "mmd_conv_gemm": (i32, [i32, vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, C.POINTER(i32), i32, i32, i32, i32, vp]),
    "mmd_gn_conv1x1": (i32, [i32, vp, i64, vp, vp, i32, i32, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "mmd_conv_gemm_stats": (i32, [i32, vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, C.POINTER(i32), i32, i32, i32, i32, vp, i64, vp]),
    "mmd_gn_conv1x1_stats": (i32, [i32, vp, i64, vp, vp, i32, i32, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_gn_conv_gemm": (i32, [i32, vp, i64, vp, vp, i32, i32, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, C.POINTER(i32), i32, i32, i32, i32, vp]),
    "mmd_tconv_weight_bytes": (i64, [i32, i32]),
    "mmd_tconv_pack": (i32, [vp, vp, i32, i32, vp]),
    "mmd_tconv": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_aconv": (i32, [vp, i64, vp, vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_tattn_weight_bytes": (i64, [i32, i32]),
    "mmd_tattn_pack": (i32, [vp, vp, vp, vp, i32, vp]),
    "mmd_tattn_block": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, f32, vp, i64, i32, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_vconv2d1d_weight_bytes": (i64, [i32]),
    "mmd_vconv2d1d_pack": (i32, [vp, vp, vp, i32, i32, vp]),
    "mmd_vconv2d1d": (i32, [vp, i64, vp, vp, i32, i32, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_zero": (i32, [vp, i64, vp]),
    "mmd_gn_finalize_stats": (i32, [vp, i64, i32, i32, i32, vp, vp, vp, i64, f32, vp, vp, vp, vp]),
    "mmd_attn_fwd": (i32, [i32, vp, i64, i32, vp, i64, i32, i32, vp, i64, i32, i32, i32, i32, i64, i32, i64, i32, i32, vp, i32, vp]),
    "mmd_attn_small_fwd": (i32, [i32, vp, i64, vp, i64, i32, i32, i32, i32, i32, i64, i64, i64, vp]),
    "mmd_resample": (i32, [i32, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "mmd_resample_stats": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp]),
    "mmd_copy2d": (i32, [vp, i64, vp, i64, i64, i64, vp]),
    "mmd_stem_conv": (i32, [i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), vp]),
    "mmd_head_conv": (i32, [i32, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), vp]),
    "mmd_head_gemm_weight_bytes": (i64, [i32]),
    "mmd_head_gemm_workspace_bytes": (i64, [i64, i32, i32]),
    "mmd_head_gemm": (i32, [vp, i64, i64, i32, vp, vp, i32, i64, i32, vp, vp, i32, vp]),
    "mmd_head_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(i32), vp]),
    "mmd_ddpm_update": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mmd_loss_workspace_bytes": (i64, [i32]),
    "mmd_loss_terms": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "mmd_conv_wgrad": (i32, [i32, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, C.POINTER(i32), i32, i32, i32, i32, vp]),
    "mmd_gn_bwd": (i32, [i32, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, vp]),
    "mmd_gn_bwd_ws0": (i32, [i32, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, i64, i64, i64, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, vp]),
    "mmd_attn_bwd": (i32, [i32, vp, i64, i32, vp, i64, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, i64, i32, i32, vp, vp, i32, i32, i32, i32,
                           i32, i64, i64, i64, i32, i32, i32, i64, i64, i64, i32, i32, i32, vp, vp]),
    "mmd_timestep_embedding": (i32, [vp, i32, i32, i32, vp, vp]),
    "mmd_attn_fwd_lse": (i32, [i32, vp, i64, i32, vp, i64, i32, i32, vp, i64, i32, i32, i32, i32, i64, i32, i64, i32, i32, vp, vp, vp]),
    "mmd_attn_bwd_mfma": (i32, [vp, i64, i32, vp, i64, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, i64, i32, i32, vp, vp, i32, i32, i32, i32,
                                i64, i32, i64, i32, i32, vp, vp]),
    "mmd_attn_small_bwd": (i32, [i32, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i64, i64, i64, vp]),
    "mmd_ddim_update": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "mmd_lincomb_t": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i64, vp]),
    "mmd_lincomb": (i32, [vp, f32, vp, f32, vp, f32, vp, i64, vp]),
    "mmd_cast": (i32, [vp, i32, vp, i32, f32, i64, vp]),
    "mmd_ddpm_update_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "mmd_abs_quantile": (i32, [vp, i32, i64, f32, vp, vp]),
    "mmd_clamp_scale": (i32, [vp, vp, f32, i32, i64, vp]),
    "mmd_dpm_err": (i32, [vp, vp, vp, f32, f32, i32, i64, vp, vp]),
    "mmd_bilinear_concat": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mmd_bilinear_concat_rows": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mmd_pack_conv_weights": (i32, [i32, vp, i32, i32, vp]),
    "mmd_pack_blocks": (i32, [i32, i32, i32]),
    "mmd_unpack_conv_grads": (i32, [vp, i32, i32, vp]),
    "mmd_loss_terms_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "mmd_silu": (i32, [i32, vp, vp, vp, i64, vp]),
    "mmd_dropout": (i32, [i32, vp, vp, f32, vp, i64, vp]),
    "mmd_mse_grad": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "mmd_adamw_step": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "mmd_q_sample": (i32, [vp, vp, vp, vp, vp, i32, i32, i64, vp]),
}
EXPORTS = tuple(_PROTOS)


class MMDError(RuntimeError):
    pass


def lib():
    """Load libmmd.so (once).  Raises MMDError when it has not been built - never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMDError(f"{LIB_PATH} not found - build it with `python mm-diffusion_amd/build.py` "
                           "(the MI355X HIP path has no CPU/torch fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise MMDError(f"{name} failed ({rc}): {lib().mmd_last_error().decode()}")


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise MMDError(f"unsupported activation dtype {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


def taps_array(taps):
    flat = [int(v) for t in taps for v in t]
    return (i32 * len(flat))(*flat), len(taps)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MMDError("the MI355X HIP path needs device tensors (got a CPU tensor); there is no CPU fallback")


class preserve_rng:
    """Keep torch's default generator of `device` out of plan building.  The per-shape tile autotuner (ops._pick_tile) fills its scratch
    inputs with N(0,1) from that generator the FIRST time a shape is seen in a process, so without this a seeded sampling loop
    (`th.manual_seed(s)` ... `p_sample_loop`, reference gd:547-550 / 453-454) would draw a different noise stream depending on whether
    the engine was already built - the reference's stream has no such consumer (measured: profiles/r05_rccl_world1_and_rng.txt)."""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.state = None

    def __enter__(self):
        if self.dev.type == "cuda":
            self.state = torch.cuda.get_rng_state(self.dev)
        return self

    def __exit__(self, *exc):
        if self.state is not None:
            torch.cuda.set_rng_state(self.state, self.dev)
        return False


# ----------------------------------------------------------------------------- lifetime of HIP handles
# Graph execs, events and streams are owned by Python objects (engines, steppers) that sit in reference cycles, so their
# finalisers run from the cyclic GC at arbitrary points - e.g. in the middle of ANOTHER engine's stream capture, where
# hipGraphExecDestroy / hipStreamDestroy are illegal.  Rule: close() / __del__ never touch the HIP runtime; they `retire` the
# handle, and `reap()` destroys retired handles at safe points only (engine construction, before a capture begins).
_retired = []
_capturing = 0


def retire(kind, handle):
    """kind: "graph" | "event" | "stream".  Safe from a finaliser (appends to a list, nothing else)."""
    if isinstance(handle, C.c_void_p):
        handle = handle.value
    if handle and _retired is not None:
        _retired.append((kind, int(handle)))


def reap():
    """Destroy retired handles.  No-op during a capture.  Work enqueued with them is drained first."""
    if _capturing or not _retired:
        return 0
    items = list(_retired)
    del _retired[:]
    torch.cuda.synchronize()
    l = lib()
    for kind in ("graph", "event", "stream"):          # execs before the events/streams their nodes were recorded with
        fn = getattr(l, "mmd_%s_destroy" % kind)
        for k, h in items:
            if k == kind:
                fn(h)
    return len(items)


class capture:
    """`with capture(stream) as c:` records the launches made on `stream` (and the streams forked from it through events) into a
    hipGraph; c.exec is the instantiated executable.  The cyclic GC is off inside (no finaliser runs mid-capture)."""

    def __init__(self, stream):
        self.stream, self.exec = stream, None

    def __enter__(self):
        global _capturing
        reap()
        self._gc = gc.isenabled()
        gc.disable()
        try:
            call("mmd_graph_begin", self.stream)
        except Exception:
            if self._gc:
                gc.enable()
            raise
        _capturing += 1
        return self

    def __exit__(self, et, ev, tb):
        global _capturing
        ex = C.c_void_p()
        try:
            rc = lib().mmd_graph_end(self.stream, C.byref(ex))
        finally:
            _capturing -= 1
            if self._gc:
                gc.enable()
        if rc == 0:
            self.exec = ex
            if et is not None:
                retire("graph", ex)
                self.exec = None
        elif et is None:
            raise MMDError(f"mmd_graph_end failed ({rc}): {lib().mmd_last_error().decode()}")
        return False


class Stream:
    """A private launch stream (hipStreamNonBlocking, created by libmmd - never one of torch's 32 pooled streams, which alias
    each other once more than 32 have been handed out) with its torch view for wait_stream() / `with torch.cuda.stream()`."""

    def __init__(self, device):
        device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(device):
            call("mmd_stream_create", C.byref(h))
        self.handle = h.value
        self.torch = torch.cuda.ExternalStream(self.handle, device=device)

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            retire("stream", h)

    __del__ = close


def plan_events(*plans):
    """The fork/join events a recorded plan owns (ops.record_sync)."""
    return [args[2] for plan in plans for fn, args, *_ in plan if fn is None]


class Staged:
    """Asynchronous host->device upload of a small per-step buffer through a RING of pinned slots.  The host runs several
    steps ahead of the GPU (a replayed step is ~15 ms of GPU time against < 1 ms of host time), so ONE pinned buffer refilled
    every step would be overwritten before the earlier hipMemcpyAsync has read it; a slot is rewritten only after the event
    recorded behind its last copy has completed (which also bounds the host's lead to `depth` steps)."""

    def __init__(self, dev, depth=8):
        self.dev = dev
        self.slots = [torch.empty(dev.shape, dtype=dev.dtype).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.k = 0

    def host(self):
        ev = self.events[self.k]
        if ev is not None:
            ev.synchronize()
        return self.slots[self.k]

    def push(self):
        k = self.k
        self.dev.copy_(self.slots[k], non_blocking=True)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record()
        self.k = (k + 1) % len(self.slots)
