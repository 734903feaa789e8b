"""Thin per-op Python wrappers over the C-ABI (one function per libmmd entry point).

Activations are 2-D torch tensors [rows, C] with stride(1) == 1 and any row stride (a column slice of a
wider buffer is fine); torch is used for device memory and the current stream only - every arithmetic
op runs in libmmd."""
import math
import os

import torch

from . import _hip as H

GN_EPS = 1e-5

# When a recorder list is installed, ops append (cfunc, args, name) instead of launching: the engine replays
# such a plan with `cfunc(*args, stream)` (every libmmd entry point takes the stream as its LAST argument).
_recorder = None
_keep = None        # recording(keep=[...]): every buffer an op allocates while recording is parked here (a plan holds raw pointers)


class recording:
    def __init__(self, plan, keep=None):
        self.plan, self.keep = plan, keep

    def __enter__(self):
        global _recorder, _keep
        self._prev, _recorder = (_recorder, _keep), self.plan
        _keep = self.keep
        return self.plan

    def __exit__(self, *a):
        global _recorder, _keep
        _recorder, _keep = self._prev


def alloc(*shape, dtype, device):
    """torch.empty for op outputs / workspaces.  While a plan is being recorded with a keep-list the buffer is parked there: the
    plan replays raw pointers, so everything it touches must outlive it (image_unet's graph-replayed forward)."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if _keep is not None and _recorder is not None:
        _keep.append(t)
    return t


cur_sid = 0      # launch stream of the ops being recorded: 0 = video/main stream, 1 = audio stream
cur_tag = ""     # which reference module the recorded launches belong to (bench: per-block roofline accounting)


def _dispatch(name, *args, meta=None):
    """meta = (kernel label, algorithmic flops, algorithmic HBM bytes) of this launch (bench roofline accounting)."""
    if _recorder is not None:
        _recorder.append((getattr(H.lib(), name), args, name, meta or (name, 0, 0), cur_sid, cur_tag))
    else:
        H.call(name, *args, H.stream_handle())


def record_sync(src, dst):
    """Plan marker: stream `dst` waits for everything recorded so far on stream `src` (event record + wait)."""
    import ctypes
    ev = ctypes.c_void_p()
    H.call("mmd_event_create", ctypes.byref(ev))
    _recorder.append((None, (src, dst, ev), "sync", None, -1, ""))


# EXPERIMENT switch (timing only, never set by the product or the tests): MMD_SKIP_SID=0 / 1 drops the launches of the video / audio
# chain from a replayed plan - the step time without one chain bounds what that chain costs the other (interference + waits)
_SKIP_SID = int(os.environ["MMD_SKIP_SID"]) if os.environ.get("MMD_SKIP_SID", "") in ("0", "1") else None
# likewise timing only: MMD_SKIP_WAIT=10 drops every "video stream waits for the audio stream" marker, =01 the opposite direction - how long
# does a chain actually stand waiting for the other one (the numbers of such a run mean nothing)
_SKIP_WAIT = os.environ.get("MMD_SKIP_WAIT", "")
if _SKIP_SID is not None or _SKIP_WAIT:
    import warnings
    warnings.warn("MMD_SKIP_SID / MMD_SKIP_WAIT are set: replayed plans DROP a launch chain or cross-stream waits - timing experiments only, "
                  "every output of this process is wrong", RuntimeWarning)
_TIMING_ONLY_OK = os.environ.get("MMD_TIMING_ONLY", "") == "1"      # the bench / tools opt in explicitly; anything else refuses the switches


def run_plan(plan, stream, aux_stream=None):
    """Replay a recorded plan.  With aux_stream the audio-stream ops run there (fork/join through events, also
    valid under stream capture); without it everything runs in recording order on `stream` (markers are no-ops)."""
    lib = H.lib()
    if (_SKIP_SID is not None or _SKIP_WAIT) and not _TIMING_ONLY_OK:
        raise H.MMDError("MMD_SKIP_SID / MMD_SKIP_WAIT (timing experiments: they drop launches / waits from every replayed plan) need "
                         "MMD_TIMING_ONLY=1 as well - refusing to replay a plan whose outputs would be silently wrong")
    streams = (stream, aux_stream if aux_stream is not None else stream)
    for fn, args, name, _, sid, _tag in plan:
        if fn is None:
            if aux_stream is not None:
                src, dst, ev = args
                if _SKIP_WAIT and _SKIP_WAIT == f"{src}{dst}":
                    continue
                if lib.mmd_event_record(ev, streams[src]) or lib.mmd_stream_wait_event(streams[dst], ev):
                    raise H.MMDError(f"sync failed: {lib.mmd_last_error().decode()}")
            continue
        if _SKIP_SID is not None and sid == _SKIP_SID:
            continue
        rc = fn(*args, streams[sid])
        if rc != 0:
            raise H.MMDError(f"{name} failed ({rc}): {lib.mmd_last_error().decode()}")


class Geom:
    """Slice geometry of a GroupNorm / short-attention instance (see include/mmd.h: mmd_gn_stats)."""
    __slots__ = ("S", "Tn", "inner", "outer_stride", "inner_stride", "tstride")

    def __init__(self, S, Tn, inner=1, outer_stride=None, inner_stride=1, tstride=1):
        self.S, self.Tn, self.inner = int(S), int(Tn), int(inner)
        self.outer_stride = int(Tn if outer_stride is None else outer_stride)
        self.inner_stride, self.tstride = int(inner_stride), int(tstride)

    @staticmethod
    def per_sample(N, rows_per_sample):
        return Geom(N, rows_per_sample, 1, rows_per_sample, 1, 1)

    @staticmethod
    def spatial(N, F, HW):
        return Geom(N * F, HW, 1, HW, 1, 1)

    @staticmethod
    def temporal(N, F, HW):
        return Geom(N * HW, F, HW, F * HW, 1, HW)

    def args(self):
        return (self.S, self.Tn, self.inner, self.outer_stride, self.inner_stride, self.tstride)


def _chk2d(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise H.MMDError(f"expected a [rows, C] tensor with unit channel stride, got {tuple(t.shape)} strides {t.stride()}")
    H.require_cuda(t)


def gn_workspace_bytes(x, geom: Geom):
    return max(8, int(H.lib().mmd_gn_workspace_bytes(H.dt_of(x), x.shape[1], geom.S, geom.Tn)))


def gn_workspace(x, geom: Geom):
    return alloc(gn_workspace_bytes(x, geom) // 8, dtype=torch.float64, device=x.device)


def gn_stats(x, gamma, beta, geom: Geom, film=None, a=None, b=None, ws=None, mr=None):
    _chk2d(x)
    C = x.shape[1]
    a = alloc(geom.S, C, dtype=torch.float32, device=x.device) if a is None else a
    b = alloc(geom.S, C, dtype=torch.float32, device=x.device) if b is None else b
    ws = gn_workspace(x, geom) if ws is None else ws
    _dispatch("mmd_gn_stats", H.dt_of(x), x.data_ptr(), x.stride(0), C, *geom.args(), gamma.data_ptr(), beta.data_ptr(),
           H.ptr(film), 0 if film is None else film.stride(0), GN_EPS, a.data_ptr(), b.data_ptr(), H.ptr(mr), ws.data_ptr(),
           meta=(f"gn_stats[S={geom.S},Tn={geom.Tn},C={C}]", 0, geom.S * geom.Tn * C * x.element_size()))
    return a, b


def gn_finalize_stats(rec, gamma, beta, geom: Geom, film=None, a=None, b=None, mr=None):
    """The fused GroupNorm affine from producer-side statistics (include/mmd.h: mmd_gn_finalize_stats): rec = fp32 view
    [rows / 64, C / 4, 2] (one record per QUAD of channels) over exactly the channels of the normalised tensor; S contiguous slices
    of Tn rows."""
    C = rec.shape[1] * 4
    if C % 128:
        raise H.MMDError("gn_finalize_stats: the normalised channels must be a multiple of 128 (groups of whole quads)")
    if geom.inner != 1 or geom.tstride != 1 or geom.outer_stride != geom.Tn or geom.Tn % 64 or rec.shape[0] * 64 != geom.S * geom.Tn:
        raise H.MMDError("gn_finalize_stats: needs contiguous slices that are multiples of 64 rows")
    a = alloc(geom.S, C, dtype=torch.float32, device=rec.device) if a is None else a
    b = alloc(geom.S, C, dtype=torch.float32, device=rec.device) if b is None else b
    _dispatch("mmd_gn_finalize_stats", rec.data_ptr(), rec.stride(0) // 2, C, geom.S, geom.Tn, gamma.data_ptr(), beta.data_ptr(),
              H.ptr(film), 0 if film is None else film.stride(0), GN_EPS, a.data_ptr(), b.data_ptr(), H.ptr(mr),
              meta=(f"gn_finalize_stats[S={geom.S},Tn={geom.Tn},C={C}]", 0, rec.shape[0] * C * 8))
    return a, b


def gn_apply(x, a, b, geom: Geom, act=True, out=None):
    _chk2d(x)
    out = alloc(x.shape, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    _dispatch("mmd_gn_apply", H.dt_of(x), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
           *geom.args(), a.data_ptr(), b.data_ptr(), 1 if act else 0,
           meta=("gn_apply", 0, 2 * x.shape[0] * x.shape[1] * x.element_size()))
    return out


def zero(t):
    """hipMemsetAsync(0) of a device tensor on the launch stream (a memset node in a captured plan)."""
    H.require_cuda(t)
    _dispatch("mmd_zero", t.data_ptr(), t.numel() * t.element_size(), meta=("zero", 0, t.numel() * t.element_size()))
    return t


def gn_small_ok(x, geom: Geom):
    return geom.Tn <= 16 and x.shape[1] % 128 == 0 and x.shape[1] // (8 if x.element_size() == 2 else 4) <= 256


def gn_small(x, gamma, beta, geom: Geom, act=False, out=None):
    """One-launch GroupNorm32(+SiLU) for short slices (include/mmd.h: mmd_gn_small) = gn_stats + gn_apply."""
    _chk2d(x)
    out = alloc(x.shape, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    _dispatch("mmd_gn_small", H.dt_of(x), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[1], *geom.args(), gamma.data_ptr(),
              beta.data_ptr(), GN_EPS, 1 if act else 0, meta=(f"gn_small[S={geom.S},Tn={geom.Tn},C={x.shape[1]}]", 0, 2 * x.shape[0] * x.shape[1] * x.element_size()))
    return out


# One-launch GroupNorm for slices of a few hundred rows (mmd_gn_group): used where a norm has NO producer-side records to finalize from
# and its (group, slice) fits a block's registers - the 400-row audio samples at ds8 (400 % 64 != 0: gn_stats there is a partial
# launch + a finalize launch, then gn_apply).  MMD_GN_GROUP=0 switches it off (A/B).
_GN_GROUP = os.environ.get("MMD_GN_GROUP", "1") != "0"


def gn_group_ok(x, geom: Geom):
    C = x.shape[1]
    return _GN_GROUP and C % 128 == 0 and C <= 2048 and geom.Tn * (C // 128) <= 4096 and x.stride(0) % 4 == 0


def gn_group(x, gamma, beta, geom: Geom, film=None, a=None, b=None, out=None, act=False, mr=None):
    """GroupNorm32(+FiLM)(+SiLU) in one launch (include/mmd.h: mmd_gn_group): the fused affine into a / b (both or neither), the
    normalised tensor into out (or neither: then a / b are allocated) - gn_stats (+ gn_apply) for short slices."""
    _chk2d(x)
    C = x.shape[1]
    if out is None and a is None:
        a = alloc(geom.S, C, dtype=torch.float32, device=x.device)
        b = alloc(geom.S, C, dtype=torch.float32, device=x.device)
    if out is not None:
        _chk2d(out)
    _dispatch("mmd_gn_group", H.dt_of(x), x.data_ptr(), x.stride(0), H.ptr(out), 0 if out is None else out.stride(0), C, *geom.args(),
              gamma.data_ptr(), beta.data_ptr(), H.ptr(film), 0 if film is None else film.stride(0), GN_EPS, 1 if act else 0,
              H.ptr(a), H.ptr(b), H.ptr(mr),
              meta=(f"gn_group[S={geom.S},Tn={geom.Tn},C={C}]", 0, (1 if out is None else 2) * geom.S * geom.Tn * C * x.element_size()))
    return (a, b) if out is None else out


def add_rowbias(x, e, rows_per_sample):
    _chk2d(x)
    _dispatch("mmd_add_rowbias", H.dt_of(x), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], rows_per_sample,
           e.data_ptr(), e.stride(0))
    return x


def colsum_slices(dy, out, rows_per_sample):
    """out[s, :] += column sums of the rows of sample s of dy (include/mmd.h: mmd_colsum_slices); out fp32 [S, C], zeroed by the caller."""
    _chk2d(dy)
    S = dy.shape[0] // rows_per_sample
    _dispatch("mmd_colsum_slices", H.dt_of(dy), dy.data_ptr(), dy.stride(0), S, rows_per_sample, dy.shape[1], out.data_ptr(), out.stride(0))
    return out


# ---- tap tables (offsets of (p0, p1, p2))
TAPS_1 = [(0, 0, 0)]
TAPS_SPATIAL = [(0, dh, dw) for dh in (-1, 0, 1) for dw in (-1, 0, 1)]      # D = (1|F, H, W)
TAPS_TEMPORAL = [(df, 0, 0) for df in (-1, 0, 1)]                           # D = (F, HW, 1)
TAPS_TEMPORAL_D1 = [(0, df, 0) for df in (-1, 0, 1)]                        # the same conv with D = (N, F, HW): the form tile 130 accepts
TAPS_3D = [(df, dh, dw) for df in (-1, 0, 1) for dh in (-1, 0, 1) for dw in (-1, 0, 1)]


def taps_audio(dilation):
    return [(-dilation, 0, 0), (0, 0, 0), (dilation, 0, 0)]                 # D = (L, 1, 1)


# ---- per-shape tile choice: measured once per (dtype, M, K, N, taps, fused-GN) on the real buffers at plan-build time
AUTOTUNE = True
_tile_cache = {}


def _pick_tile(key, launch, M, Cout, candidates=(64, 128, 129), out=None, scratch=()):
    """launch(tile) enqueues the GEMM on the current stream.  Returns the fastest of `candidates`
    (64 / 128 = register-staged tiles, 129 = 128x128 direct-to-LDS main loop).

    A candidate is only eligible after its output on the real buffers agrees with the first candidate's (tile 64, the reference
    variant): bitwise for 64 / 128 / 129 (same K order, same epilogue), within rounding for tile 130 (chunk-major K order).  A
    variant that disagrees is a kernel bug and raises - timing alone never admits a kernel into a captured graph.  `scratch` =
    the input buffers (plan-time garbage) that are filled with N(0,1) first so the comparison sees finite, representative data."""
    default = (129 if 129 in candidates else 128) if ((M + 127) // 128) * ((Cout + 127) // 128) >= 320 else 64
    if default not in candidates:
        default = candidates[0]
    if not AUTOTUNE or _recorder is None:
        return default
    key = (key, tuple(candidates))      # a shape tuned over (64, 128, 129) must not answer for a launch restricted to (128, 129)
    if key in _tile_cache:
        return _tile_cache[key]
    import ctypes
    verify = out is not None and not any(t is not None and t.untyped_storage().data_ptr() == out.untyped_storage().data_ptr() for t in scratch)
    saved = []
    if verify:
        for t in scratch:
            if t is not None:
                saved.append((t, t.clone()))      # a plan recorded over a persistent buffer (sampler state, static inputs) gets it back
                t.normal_()
    best, best_ms, ref = default, None, None
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    for tile in candidates:
        launch(tile)                      # warm (function attributes, caches)
        if verify:
            if ref is None:
                ref = out.clone()
            elif tile == 130:
                err = float((out.float() - ref.float()).norm() / ref.float().norm().clamp_min(1e-30))
                if not err < (1e-2 if out.element_size() == 2 else 1e-5):
                    raise H.MMDError(f"conv_gemm tile 130 disagrees with tile {candidates[0]} on {key}: rel-L2 {err:.3e}")
            elif not torch.equal(out.view(torch.int16 if out.element_size() == 2 else torch.int32),
                                 ref.view(torch.int16 if ref.element_size() == 2 else torch.int32)):
                raise H.MMDError(f"GEMM tile {tile} is not bitwise equal to tile {candidates[0]} on {key}")
        t_min = None
        for _ in range(2):               # the better of two rounds of five launches: three launches once picked a 30 % slower tile
            H.call("mmd_event_record", ev[0], st)
            for _ in range(5):
                launch(tile)
            H.call("mmd_event_record", ev[1], st)
            ms = ctypes.c_float()
            H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
            t_min = ms.value if t_min is None else min(t_min, ms.value)
        if best_ms is None or t_min < best_ms:
            best, best_ms = tile, t_min
    for e in ev:
        H.lib().mmd_event_destroy(e)
    for t, keep in saved:
        t.copy_(keep)
    _tile_cache[key] = best
    return best


# tile 130 (halo-tile main loop for 3x3 / temporal-k3 convs).  Measured on MI355X (tools/gemm_bench.py, batch 4): 3x3 ds1 128->128
# 119.7 -> 104.9 us, 256->128 190 -> 175, ds2 256->256 91 -> 86, ds4 tie.  Its K order is chunk-major, so it is NOT bitwise equal to
# tiles 64 / 128 / 129: a timing-based choice would let the batch-4 and the batch-1 plan of the same layer differ in the last bit
# (tests: batch rows == batch-1 runs, bitwise).  It is therefore chosen by the LAYER GEOMETRY alone (halo_tile_pinned), never by the
# autotuner, unless MMD_GEMM_HALO=1 asks for the old experiment (candidate everywhere it is legal) or =0 switches it off.
_HALO_MODE = os.environ.get("MMD_GEMM_HALO", "pin")
HALO_CANDIDATE = _HALO_MODE == "1"


def halo_tile_ok(x, taps, dims):
    """Shapes conv_gemm tile 130 accepts: taps inside the (D1, D2) plane with |offset| <= 1, D1 % 8 == 0, D2 % 16 == 0, full frames,
    Cin a multiple of one 128-byte K step."""
    M, Cin = x.shape
    return (len(taps) > 1 and all(t[0] == 0 and abs(t[1]) <= 1 and abs(t[2]) <= 1 for t in taps) and dims[1] % 8 == 0
            and dims[2] % 16 == 0 and dims[0] * dims[1] * dims[2] == M and Cin % (128 // x.element_size()) == 0)


# tile 133 = the halo-tile main loop on 16 x 16 patches (8 waves, one block per CU, three-slot weight ring): same K order as tile 130,
# bitwise the same output, so wherever tile 130 is pinned and the frame sides are multiples of 16 the faster of the two may run
# (MMD_HALO16=0: always tile 130).
_HALO16 = os.environ.get("MMD_HALO16", "1") != "0"


def halo_tile_code(x, taps, dims):
    """130 or 133 for a launch tile 130 accepts.  Measured (tools/halo_bench.py, MI355X): the one-block-per-CU tile 133 wins from four
    channel chunks on (ds1 256->128: 193 -> 164 us, ds2 640->256: 200 -> 166 us); with two chunks its prologue and epilogue have no
    co-resident block to hide behind (ds1 128->128: 103 vs 110 us)."""
    return 133 if (_HALO16 and x.dtype == torch.bfloat16 and len(taps) == 9 and dims[1] % 16 == 0 and dims[2] % 16 == 0
                   and x.shape[1] >= 256 and tuple(tuple(t) for t in taps) == tuple(TAPS_SPATIAL)) else 130


def _stats_args(stats, M, Cout):
    """stats: fp32 view [M / 64, Cout / 4, 2] (a quad-column slice of the output's record buffer) -> (pointer, row stride in float2)."""
    if (stats.dtype != torch.float32 or Cout % 4 or tuple(stats.shape) != (M // 64, Cout // 4, 2) or M % 64 or stats.stride(1) != 2
            or stats.stride(2) != 1):
        raise H.MMDError(f"GEMM output statistics: expected an fp32 [M/64, Cout/4, 2] view, got {tuple(stats.shape)} strides {stats.stride()}")
    return stats.data_ptr(), stats.stride(0) // 2


# frames of >= 256 pixels since round 3 (was 1024): after the issue-side diet tile 133 ties the direct-to-LDS loop on the ds4 level
# (16 x 16 frames, 48.1 vs 48.2 us) and brings the fused input GroupNorm with it (one launch and one pass fewer per ResBlock)
# 1024: ds1 / ds2.  Pinning the ds4 frames (256 pixels) too was measured a loss in round 3: tile 133 runs those 3x3 convs in 67 us
# against 48.6 us on the direct-to-LDS tile with descriptor addressing, more than the fused GroupNorm apply gives back.
_HALO_MIN_PIXELS = int(os.environ.get("MMD_HALO_MIN_PIXELS", "1024"))


def halo_tile_pinned(x, taps, dims):
    """The layers that always run on a halo tile (130, or 133 where halo_tile_code says so): bf16 spatial 3x3 convs on frames of at
    least _HALO_MIN_PIXELS pixels (default 1024: ds1 / ds2 of the base model; MMD_HALO_MIN_PIXELS) - a property of the layer,
    independent of the batch size."""
    return (_HALO_MODE == "pin" and x.dtype == torch.bfloat16 and len(taps) == 9 and dims[1] * dims[2] >= _HALO_MIN_PIXELS
            and halo_tile_ok(x, taps, dims))


# tile 131 (row-strip main loop, mmd_gemm.hip: a wave's rows stationary in registers as MFMA operands, GroupNorm applied once per
# strip, the weights streamed through LDS) for the bf16 convs whose whole K = ntaps * Cin is 128 ... 512: the 1x1 convs at 128-512
# channels and the k=3 temporal / audio convs at 128 channels.  Its output is bitwise equal to tiles 64 / 128 / 129, its output
# STATISTICS are folded in its own order, so - like tile 130 - it is chosen by the layer's channel geometry alone, never by timing:
# a layer runs the same kernel at every batch size.  MMD_GEMM_STRIP=0 switches it off (A/B); =base restricts it to the first
# validated subset (1x1 convs, Cin 128 / 256 / 384, statistics only up to 256 channels).
_STRIP_MODE = os.environ.get("MMD_GEMM_STRIP", "pin")


def strip_tile_ok(x, Cout, taps=TAPS_1, stats=None, geom=None, base=False):
    """Launches conv_gemm tile 131 accepts: bf16, K = ntaps * Cin in {128, 256} (two row fragments per wave, 64-channel chunks) or
    {384, 512} (one fragment, 32-channel chunks), Cin a multiple of 64; output statistics need M % 64 == 0; fused GroupNorm (1x1 convs
    only) needs contiguous slices of at least one block of rows (256 / 128)."""
    M, Cin = x.shape
    K = len(taps) * Cin
    if x.dtype != torch.bfloat16 or K not in (128, 256, 384, 512) or Cin % 64:
        return False
    if len(taps) == 1 and tuple(taps[0]) != (0, 0, 0):
        return False
    rf = 2 if K <= 256 else 1
    if Cout % (32 * rf) or Cout > 2048 or (stats is not None and M % 64):      # <= 2048 output channels per block (its LDS bias table)
        return False
    if base and (len(taps) != 1 or K == 512 or (stats is not None and rf != 2)):
        return False
    if geom is not None and not (len(taps) == 1 and geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn
                                 and geom.Tn >= 128 * rf and geom.S * geom.Tn == M):
        return False
    return True


def strip_tile_pinned(x, Cout, taps=TAPS_1, stats=None, geom=None):
    if _STRIP_MODE == "0":
        return False
    return strip_tile_ok(x, Cout, taps, stats, geom, base=_STRIP_MODE == "base")


def _tile_name(tile):
    return {129: "128glds", 130: "128halo", 131: "strip", 132: "128ring", 133: "256halo"}.get(tile, tile)


# tile 132 (mmd_gemm.hip: the direct-to-LDS loop with a four-slot LDS ring, three K steps of DMA in flight, one block per CU) is
# bitwise equal to tiles 64 / 128 / 129 (statistics included: the 128-row epilogue), so it simply joins the autotune candidates -
# where it can win: launches with at most ~1.5 tiles per CU (nothing co-resident hides a block's L2 round trips) and >= 4 K steps.
_RING_MODE = os.environ.get("MMD_GEMM_RING", "1")


def ring_tile_candidate(x, Cout, ntaps):
    M, Cin = x.shape
    kstep = 128 // x.element_size()
    return (_RING_MODE != "0" and Cin % kstep == 0 and (Cin * ntaps) // kstep >= 4
            and ((M + 127) // 128) * ((Cout + 127) // 128) <= 384)


def conv_gemm(x, w, bias, taps=TAPS_1, dims=(1, 1, 1), residual=None, out=None, tile=0, stats=None):
    """x [M, Cin]; w packed [Cout, ntaps*Cin] in x.dtype; bias fp32 [Cout] or None.  stats (optional): record view that receives
    the GroupNorm statistics of the output (include/mmd.h: mmd_conv_gemm_stats)."""
    _chk2d(x)
    M, Cin = x.shape
    Cout = w.shape[0]
    if w.dtype != x.dtype or w.shape[1] != len(taps) * Cin or not w.is_contiguous():
        raise H.MMDError(f"conv_gemm: weight {tuple(w.shape)} {w.dtype} does not match input {tuple(x.shape)} {x.dtype} x {len(taps)} taps")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    arr, nt = H.taps_array(taps)
    es = x.element_size()
    base = (H.dt_of(x), x.data_ptr(), x.stride(0), w.data_ptr(), H.ptr(bias), H.ptr(residual),
            0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(0), M, Cout, Cin, nt, arr,
            int(dims[0]), int(dims[1]), int(dims[2]))
    if tile == 0 and stats is None and halo_tile_pinned(x, taps, dims):
        tile = halo_tile_code(x, taps, dims)
    if tile == 0 and strip_tile_pinned(x, Cout, taps, stats):
        tile = 131
    if tile == 0:
        # statistics-emitting launches stay inside the 128-row tile family: the per-record sums are folded in an order that depends on
        # the tile's thread layout (128 / 129 share it, 64 does not), and the choice must not move the last bit of the statistics
        # when the batch size changes the autotuner's verdict
        cands = (128, 129) if stats is not None else (64, 128, 129) + ((130,) if HALO_CANDIDATE and halo_tile_ok(x, taps, dims) else ())
        if ring_tile_candidate(x, Cout, nt):
            cands = cands + (132,)
        tile = _pick_tile((es, M, Cin, nt, Cout, residual is not None, False, tuple(dims) if 130 in cands else None),
                          lambda t: H.call("mmd_conv_gemm", *base, t, H.stream_handle()), M, Cout, cands, out=out, scratch=(x, residual))
    flops = 2 * M * Cout * Cin * nt
    nbytes = es * (M * Cin + M * Cout * (2 if residual is not None else 1) + Cout * Cin * nt) + 4 * Cout
    label = f"conv_gemm<{'bf16' if es == 2 else 'f32'},{_tile_name(tile)}>[M={M},K={Cin * nt},N={Cout}]"
    if stats is not None:
        _dispatch("mmd_conv_gemm_stats", *base, tile, *_stats_args(stats, M, Cout), meta=(label, flops, nbytes))
    else:
        _dispatch("mmd_conv_gemm", *base, tile, meta=(label, flops, nbytes))
    return out


# the threshold of the fuse / unfuse rule below.  It was the strip kernel's own split target until round 5; the kernel now splits for ONE block per
# CU (256, mmd_gemm.hip: MMD_STRIP_BLOCKS) and the rule kept its 448 (following the kernel - MMD_STRIP_BLOCKS_RULE=256, more norms fused -
# measured 10.98 against 10.94 ms: no gain)
_STRIP_BLOCKS_RULE = int(os.environ.get("MMD_STRIP_BLOCKS_RULE", "448"))


def strip_column_split(M, K, Cout):
    """A COST RULE, not the split the kernel launches: the column split the fuse / unfuse rule of gn_fusable reasons with - the smallest
    divisor of the chunk count that gives the chip >= _STRIP_BLOCKS_RULE (448) blocks, at most 16.  The kernel's own split
    (mmd_gemm.hip: launch_conv1x1_strip_mode) aims for 256 blocks since round 5; the rule kept its constant so that the fuse / unfuse
    decisions - and with them the plan the fixtures pin - did not move with a launch-width tuning."""
    rf, cc = (2, 64) if K <= 256 else (1, 32)
    rowblocks, nch, n = -(-M // (128 * rf)), Cout // cc, 1
    for d in range(1, min(nch, 16) + 1):
        if nch % d == 0:
            n = d
            if rowblocks * d >= _STRIP_BLOCKS_RULE:
                break
    return n


def gn_fusable(geom: Geom, Cin, Cout, x=None, stats=None, act=False):
    """Whether GroupNorm can ride in the 1x1 GEMM: in the tiled loader (contiguous slices of >= 128 rows, narrow K and N: every
    column tile redoes the normalisation) or, given the input x (and whether the launch will emit output statistics), in the
    row-strip kernel (normalises once per strip: any N).

    A layer whose GEMM runs on the strip kernel never goes back to the tiled loader (its statistics are folded in another order);
    what may depend on M is only whether the normalisation is FUSED: every block of a strip's column split redoes it, and with SiLU
    that is ~12 VALU instructions per element, so when few rows force a deep split (M = 4096 at 512 channels: 16 ranges, 34 us fused
    against 5 + 17 us) gn_apply + the plain strip GEMM is the faster of two bitwise-equal paths (tests/test_strip_gpu.py)."""
    if x is not None and strip_tile_pinned(x, Cout, stats=stats):
        if not strip_tile_pinned(x, Cout, stats=stats, geom=geom):
            return False                                   # slices the strip cannot fuse: gn_apply + strip GEMM
        return not (act and strip_column_split(x.shape[0], Cin, Cout) > 2)
    return (geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn and geom.Tn >= 128 and Cin <= 256
            and (Cout + 127) // 128 <= 2)


def gn_conv1x1(x, a, b, geom: Geom, act, w, bias, residual=None, out=None, tile=0, stats=None):
    """1x1 conv of GroupNorm'd rows with the normalisation fused into the GEMM loader (include/mmd.h: mmd_gn_conv1x1)."""
    _chk2d(x)
    M, Cin = x.shape
    Cout = w.shape[0]
    if w.dtype != x.dtype or w.shape[1] != Cin or not w.is_contiguous():
        raise H.MMDError(f"gn_conv1x1: weight {tuple(w.shape)} {w.dtype} does not match input {tuple(x.shape)} {x.dtype}")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    es = x.element_size()
    # capability, not preference: an explicit tile is checked against what THAT main loop can do (gn_fusable is the caller's cost rule)
    tiled_ok = (geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn and geom.Tn >= 128 and Cin <= 256
                and (Cout + 127) // 128 <= 2)
    strip_ok = strip_tile_ok(x, Cout, stats=stats, geom=geom)
    if not ((tile == 131 and strip_ok) or (tile in (64, 128) and tiled_ok) or (tile == 0 and (tiled_ok or strip_ok))):
        raise H.MMDError("gn_conv1x1: needs contiguous slices of >= 128 rows, Cin <= 256 (use gn_apply + conv_gemm otherwise)")
    base = (H.dt_of(x), x.data_ptr(), x.stride(0), a.data_ptr(), b.data_ptr(), 1 if act else 0, geom.S, geom.Tn,
            w.data_ptr(), H.ptr(bias), H.ptr(residual),
            0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(0), M, Cout, Cin)
    if tile == 0 and strip_tile_pinned(x, Cout, stats=stats, geom=geom):
        tile = 131                                         # (callers that follow gn_fusable only get here when the fusion pays)
    if tile == 0 and not tiled_ok:
        raise H.MMDError("gn_conv1x1: the row-strip kernel is switched off (MMD_GEMM_STRIP) and the tiled loader cannot take this launch")
    if tile == 0:
        tile = _pick_tile((es, M, Cin, 1, Cout, residual is not None, True),
                          lambda t: H.call("mmd_gn_conv1x1", *base, t, H.stream_handle()), M, Cout,
                          candidates=(128,) if stats is not None else (64, 128), out=out, scratch=(x, residual, a, b))
    nbytes = es * (M * Cin + M * Cout * (2 if residual is not None else 1) + Cout * Cin) + 4 * Cout
    meta = (f"gn_conv1x1<{'bf16' if es == 2 else 'f32'},{_tile_name(tile)}>[M={M},K={Cin},N={Cout}]", 2 * M * Cout * Cin, nbytes)
    if stats is not None:
        _dispatch("mmd_gn_conv1x1_stats", *base, tile, *_stats_args(stats, M, Cout), meta=meta)
    else:
        _dispatch("mmd_gn_conv1x1", *base, tile, meta=meta)
    return out


# GroupNorm(+FiLM)(+SiLU) of the INPUT of a 3x3 conv applied to the staged halo tile in LDS (mmd_gn_conv_gemm): the gn_apply pass in
# front of the ResBlock in-convs disappears wherever the conv runs on tile 130.  Bitwise equal to gn_apply + conv_gemm(tile 130), so
# the switch (MMD_HALO_GN=0: separate pass) is a pure speed choice.
_HALO_GN = os.environ.get("MMD_HALO_GN", "1") != "0"


def halo_gn_ok(x, taps, dims, geom: Geom):
    """Launches mmd_gn_conv_gemm accepts: what tile 130 is pinned on (bf16 3x3 convs on frames of >= 1024 pixels) with per-sample
    slices of whole frames."""
    return (_HALO_GN and halo_tile_pinned(x, taps, dims) and geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn
            and geom.S * geom.Tn == x.shape[0] and geom.Tn % (dims[1] * dims[2]) == 0)


def gn_conv_gemm(x, a, b, geom: Geom, act, w, bias, taps, dims, residual=None, out=None, tile=0):
    """3x3 conv of GroupNorm'd rows with the normalisation applied to the staged halo (include/mmd.h: mmd_gn_conv_gemm, tiles 130 / 133)."""
    _chk2d(x)
    M, Cin = x.shape
    Cout = w.shape[0]
    if w.dtype != x.dtype or w.shape[1] != len(taps) * Cin or not w.is_contiguous():
        raise H.MMDError(f"gn_conv_gemm: weight {tuple(w.shape)} {w.dtype} does not match input {tuple(x.shape)} {x.dtype} x {len(taps)} taps")
    if not (halo_tile_ok(x, taps, dims) and len(taps) == 9 and x.dtype == torch.bfloat16 and geom.S * geom.Tn == M
            and geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn and geom.Tn % (dims[1] * dims[2]) == 0):
        raise H.MMDError("gn_conv_gemm: needs a bf16 3x3 conv tile 130 accepts and contiguous slices of whole frames")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    arr, nt = H.taps_array(taps)
    es = x.element_size()
    flops = 2 * M * Cout * Cin * nt
    nbytes = es * (M * Cin + M * Cout * (2 if residual is not None else 1) + Cout * Cin * nt) + 4 * Cout
    _dispatch("mmd_gn_conv_gemm", H.dt_of(x), x.data_ptr(), x.stride(0), a.data_ptr(), b.data_ptr(), 1 if act else 0, geom.S, geom.Tn,
              w.data_ptr(), H.ptr(bias), H.ptr(residual), 0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(0),
              M, Cout, Cin, nt, arr, int(dims[0]), int(dims[1]), int(dims[2]), tile or halo_tile_code(x, taps, dims),
              meta=(f"gn_conv_gemm<bf16,{_tile_name(tile or halo_tile_code(x, taps, dims))}>[M={M},K={Cin * nt},N={Cout}]", flops, nbytes))
    return out


# ---- VideoConv '2d+1d' in one launch (mmd_vconv2d1d): spatial 3x3 + temporal k=3 with the intermediate in LDS, the in_layers norm +
# SiLU applied to the staged halo and the output statistics in the epilogue.  Its spatial K order is its own (32-channel chunks), so -
# like the halo tiles and the strip - it is chosen by the LAYER GEOMETRY alone: bf16, 16 frames, 128 output channels (the ds1 level of
# the base model), Cin a multiple of 32, frame sides multiples of 4.  MMD_VCONV_FUSED=0: the two-launch path (A/B).
_VCONV_FUSED = os.environ.get("MMD_VCONV_FUSED", "1") != "0"


def vconv_shape_ok(x, Cout, N, F, Hh, Ww):
    """Launches mmd_vconv2d1d accepts."""
    return (x.dtype == torch.bfloat16 and F == 16 and Cout == 128 and x.shape[1] % 32 == 0 and Hh % 4 == 0 and Ww % 4 == 0
            and x.shape[0] == N * F * Hh * Ww and (16 * Hh * Ww * x.stride(0) + x.shape[1]) * 2 < 2 ** 31)


def _ranges_overlap(x, y):
    """The C side's X / Y check of mmd_vconv2d1d: whole address ranges [first byte, last byte] of the two row-strided views."""
    x0, y0 = x.data_ptr(), y.data_ptr()
    x1 = x0 + ((x.shape[0] - 1) * x.stride(0) + x.shape[1]) * x.element_size()
    y1 = y0 + ((y.shape[0] - 1) * y.stride(0) + y.shape[1]) * y.element_size()
    return not (x1 <= y0 or y1 <= x0)


def vconv_fused_ok(x, Cout, N, F, Hh, Ww, out=None):
    """The layers that always run on the fused kernel (a property of the layer, independent of the batch size).  out: the tensor the
    engine would write - a column slice of the buffer x lives in is refused here (the caller falls back to the two-launch path)
    instead of at launch time."""
    return _VCONV_FUSED and vconv_shape_ok(x, Cout, N, F, Hh, Ww) and (out is None or not _ranges_overlap(x, out))


def vconv_pack(ws, wt):
    """Packed spatial [128, 9 * Cin] and temporal [128, 384] bf16 GEMM matrices -> the kernel's weight image (mmd_vconv2d1d_pack)."""
    H.require_cuda(ws, wt)
    Cout, Cin = ws.shape[0], ws.shape[1] // 9
    if ws.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16 or tuple(wt.shape) != (Cout, 3 * Cout) or not ws.is_contiguous() or not wt.is_contiguous():
        raise H.MMDError(f"vconv_pack: expected contiguous bf16 [Cout, 9 Cin] / [Cout, 3 Cout], got {tuple(ws.shape)} / {tuple(wt.shape)}")
    out = torch.empty(H.lib().mmd_vconv2d1d_weight_bytes(Cin) // 2, dtype=torch.bfloat16, device=ws.device)
    H.call("mmd_vconv2d1d_pack", ws.data_ptr(), wt.data_ptr(), out.data_ptr(), Cin, Cout, H.stream_handle())
    return out


def vconv2d1d(x, wf, bias_s, bias_t, N, F, Hh, Ww, a=None, b=None, geom=None, act=True, out=None, stats=None):
    """x [N*F*H*W, Cin] bf16 -> [.., 128]: temporal_k3(spatial_3x3(act(x * a + b))) (include/mmd.h: mmd_vconv2d1d).  a / b [S, Cin]:
    the fused affine of the input norm over geom's slices (whole samples); stats: the output's record view [M / 64, 32, 2]."""
    _chk2d(x)
    M, Cin = x.shape
    Cout = 128
    if not vconv_shape_ok(x, Cout, N, F, Hh, Ww):
        raise H.MMDError(f"vconv2d1d: unsupported launch (x {tuple(x.shape)} {x.dtype}, N={N} F={F} H={Hh} W={Ww})")
    if (a is None) != (b is None) or (a is not None and (geom is None or geom.inner != 1 or geom.tstride != 1 or geom.outer_stride != geom.Tn
                                                         or geom.S * geom.Tn != M or geom.Tn % (F * Hh * Ww))):
        raise H.MMDError("vconv2d1d: the fused input norm needs contiguous slices of whole samples")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    if _ranges_overlap(x, out) or tuple(out.shape) != (M, Cout) or out.dtype != x.dtype:
        raise H.MMDError("vconv2d1d: the output must be a bf16 [M, 128] tensor whose address range does not overlap the input's (the C side "
                         "compares whole ranges: column slices of one buffer count as overlapping)")
    if wf.numel() * wf.element_size() != H.lib().mmd_vconv2d1d_weight_bytes(Cin):
        raise H.MMDError("vconv2d1d: the weight image does not match Cin (pack it with vconv_pack)")
    sp, sld = (None, 0) if stats is None else _stats_args(stats, M, Cout)
    flops = 2 * M * Cout * (9 * Cin + 3 * Cout)
    nbytes = 2 * (M * Cin + M * Cout + Cout * (9 * Cin + 3 * Cout)) + 8 * Cout
    _dispatch("mmd_vconv2d1d", x.data_ptr(), x.stride(0), H.ptr(a), H.ptr(b), 1 if act else 0, 0 if geom is None else geom.S,
              0 if geom is None else geom.Tn, wf.data_ptr(), H.ptr(bias_s), H.ptr(bias_t), out.data_ptr(), out.stride(0), N, F, Hh, Ww, Cin, Cout,
              sp, sld, meta=(f"vconv2d1d<bf16{',gn' if a is not None else ''}>[M={M},Cin={Cin},N={Cout}]", flops, nbytes))
    return out


# The temporal k = 3 conv of VideoConv with stationary activations (include/mmd.h: mmd_tconv): per-pixel row order, tap shift = DPP lane
# shift.  Bitwise equal to conv_gemm with TAPS_TEMPORAL, so the switch (MMD_TCONV=0: the tiled / strip GEMM) is a pure speed choice -
# except for the ORDER of the statistics records (the kernel's own inside a sample), which the engine accounts for (perm_unit).
_TCONV = os.environ.get("MMD_TCONV", "1") != "0"


def tconv_shape_ok(x, Cout, N, F, HW):
    return (x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] in (256, 384, 512) and Cout % 64 == 0 and 0 < Cout <= 512 and F == 16
            and HW % 8 == 0 and x.shape[0] == N * F * HW and x.stride(1) == 1 and x.stride(0) % 8 == 0)


def tconv_ok(x, Cout, N, F, HW):
    return _TCONV and tconv_shape_ok(x, Cout, N, F, HW)


def tconv_pack(w):
    """The packed temporal GEMM matrix [Cout, 3 * Cin] (bf16, K = tap * Cin + ci) -> the kernel's weight image (mmd_tconv_pack)."""
    H.require_cuda(w)
    Cout, Cin = w.shape[0], w.shape[1] // 3
    if w.dtype != torch.bfloat16 or w.dim() != 2 or w.shape[1] != 3 * Cin or not w.is_contiguous():
        raise H.MMDError(f"tconv_pack: expected a contiguous bf16 [Cout, 3 Cin] matrix, got {tuple(w.shape)} {w.dtype}")
    out = torch.empty(H.lib().mmd_tconv_weight_bytes(Cin, Cout) // 2, dtype=torch.bfloat16, device=w.device)
    H.call("mmd_tconv_pack", w.data_ptr(), out.data_ptr(), Cin, Cout, H.stream_handle())
    return out


def tconv(x, wf, bias, Cout, N, F, HW, out=None, stats=None):
    """x [N*F*HW, Cin] bf16 -> [.., Cout]: the k = 3 conv along the frames of every pixel (include/mmd.h: mmd_tconv).  stats: the
    output's record view [M / 64, Cout / 4, 2] (records in the kernel's own row order inside a sample)."""
    _chk2d(x)
    M, Cin = x.shape
    if not tconv_shape_ok(x, Cout, N, F, HW):
        raise H.MMDError(f"tconv: unsupported launch (x {tuple(x.shape)} {x.dtype}, Cout={Cout} N={N} F={F} HW={HW})")
    if wf.numel() * 2 != H.lib().mmd_tconv_weight_bytes(Cin, Cout):
        raise H.MMDError("tconv: the weight image does not match Cin / Cout")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    if out.data_ptr() == x.data_ptr():
        raise H.MMDError("tconv: in-place is not supported")
    sp, sld = (None, 0) if stats is None else _stats_args(stats, M, Cout)
    _dispatch("mmd_tconv", x.data_ptr(), x.stride(0), wf.data_ptr(), H.ptr(bias), out.data_ptr(), out.stride(0), N, F, HW, Cin, Cout, sp, sld,
              meta=(f"tconv<bf16>[M={M},K={3 * Cin},N={Cout}]", 2 * M * Cout * 3 * Cin, 2 * (M * Cin + M * Cout + Cout * 3 * Cin) + 4 * Cout))
    return out


# The audio in_layers of a ResBlock in one launch (include/mmd.h: mmd_aconv): GroupNorm + SiLU + dilated k = 3 conv, rows stationary, K
# streamed.  Bitwise equal to gn_apply + conv_gemm with taps_audio(dilation), so the switch (MMD_ACONV=0: the two launches) is a pure
# speed choice - which is why it may depend on an environment variable at all.
_ACONV = os.environ.get("MMD_ACONV", "1") != "0"
# Where the engine uses it: the kernel redoes the normalisation (~8 VALU instructions per element with SiLU) for each of the three taps
# and each column range of a row block, so it only pays where the column split is shallow - measured (profiles/r06_aconv_bench.txt, batch
# 4): 25600 x 128 -> 128 36.8 vs 41.6 us, 6400 x 256 -> 256 35.1 vs 36.6 us, but 6400 x 640 -> 256 75 vs 50, 1600 x 896 -> 384 84 vs 59,
# 400 x 1024 -> 512 63 vs 41 us (the same lesson as the row-strip kernel's fused norm in round 2).  Input widths up to MMD_ACONV_MAXCIN.
_ACONV_MAXCIN = int(os.environ.get("MMD_ACONV_MAXCIN", "256"))


def aconv_shape_ok(x, Cout, N, L):
    return (x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[0] == N * L and L >= 128 and x.shape[1] % 64 == 0 and 64 <= x.shape[1] <= 2048
            and Cout % 64 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def aconv_ok(x, Cout, N, L):
    return _ACONV and x.shape[1] <= _ACONV_MAXCIN and aconv_shape_ok(x, Cout, N, L)


def aconv(x, a, b, w, bias, N, L, dilation, act=True, out=None, stats=None):
    """x [N*L, Cin] bf16, a / b [N, Cin] fused GroupNorm affine, w [Cout, 3 Cin] packed conv weight -> [N*L, Cout] (mmd_aconv)."""
    _chk2d(x)
    M, Cin = x.shape
    Cout = w.shape[0]
    if not aconv_shape_ok(x, Cout, N, L) or w.shape[1] != 3 * Cin or w.dtype != torch.bfloat16 or not w.is_contiguous():
        raise H.MMDError(f"aconv: unsupported launch (x {tuple(x.shape)} {x.dtype}, w {tuple(w.shape)} {w.dtype}, N={N} L={L})")
    if tuple(a.shape) != (N, Cin) or tuple(b.shape) != (N, Cin) or a.dtype != torch.float32 or b.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()):
        raise H.MMDError(f"aconv: the fused affine must be two contiguous fp32 [N, Cin] tensors (got {tuple(a.shape)}, {tuple(b.shape)})")
    out = alloc(M, Cout, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    sp, sld = (None, 0) if stats is None else _stats_args(stats, M, Cout)
    _dispatch("mmd_aconv", x.data_ptr(), x.stride(0), w.data_ptr(), H.ptr(bias), a.data_ptr(), b.data_ptr(), 1 if act else 0, out.data_ptr(),
              out.stride(0), M, L, Cin, Cout, int(dilation), sp, sld,
              meta=(f"aconv<bf16>[M={M},K={3 * Cin},N={Cout}]", 2 * M * Cout * 3 * Cin, 2 * (M * Cin + M * Cout + Cout * 3 * Cin) + 4 * Cout))
    return out


# The temporal-attention block in one launch (include/mmd.h: mmd_tattn_block): GroupNorm over a pixel's frames, qkv, attention over the
# frames, proj_out and the residual - instead of gn_small + qkv GEMM + attn_small + proj_out GEMM.  Built for 256 channels (the ds2
# level), 4 heads, 16 frames; like every kernel choice it depends on the layer's geometry only.  (The 384 / 512-channel instances of
# round 4 - ds4 neutral in the step, ds8 74 vs 44 us - were removed in round 5.)  MMD_TATTN_FUSED=0: the four-launch path (A/B).
_TATTN_FUSED = os.environ.get("MMD_TATTN_FUSED", "1") != "0"
# the spatial block's proj_out + residual as the front stage of the same launch (MMD_TATTN_PRE=0: its own strip GEMM; A/B)
_TATTN_PRE = os.environ.get("MMD_TATTN_PRE", "1") != "0"


def tattn_shape_ok(x, heads, N, F, HW):
    return (x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and heads == 4 and F == 16 and HW % 8 == 0
            and x.shape[0] == N * F * HW and x.stride(1) == 1 and x.stride(0) % 8 == 0)


def tattn_fused_ok(x, heads, N, F, HW):
    return _TATTN_FUSED and tattn_shape_ok(x, heads, N, F, HW)


def tattn_pack(wqkv, wproj, wpre=None):
    """qkv weight [3 C, C] and proj_out weight [C, C] of the temporal block - and, for the front stage, the proj_out weight [C, C] of the
    spatial block before it - (bf16, contiguous GEMM matrices) -> the kernel's weight image (mmd_tattn_pack)."""
    H.require_cuda(wqkv, wproj)
    C = wproj.shape[0]
    mats = [(wqkv, (3 * C, C)), (wproj, (C, C))] + ([(wpre, (C, C))] if wpre is not None else [])
    for w, shape in mats:
        if C != 256 or w.dtype != torch.bfloat16 or tuple(w.shape) != shape or not w.is_contiguous():
            raise H.MMDError(f"tattn_pack: expected contiguous bf16 {shape} with C = 256, got {tuple(w.shape)} {w.dtype}")
    out = torch.empty(H.lib().mmd_tattn_weight_bytes(C, 0 if wpre is None else 1) // 2, dtype=torch.bfloat16, device=wqkv.device)
    H.call("mmd_tattn_pack", H.ptr(wpre), wqkv.data_ptr(), wproj.data_ptr(), out.data_ptr(), C, H.stream_handle())
    return out


def tattn_block(x, wf, bias_qkv, bias_proj, gamma, beta, heads, N, F, HW, out=None, stats=None, pre=None):
    """x [N*F*HW, C] bf16 (C = 256) -> x + proj_out(temporal attention(qkv(GroupNorm32(x)))) (include/mmd.h:
    mmd_tattn_block).  stats: the output's record view [M / 64, C / 4, 2] (records in the kernel's own row order inside a sample).  pre = (att, bias_pre, mid): the
    front stage - the block's input is x + att Wpre^T + bias_pre (wf packed with wpre), written to the scratch `mid`."""
    _chk2d(x)
    M, C = x.shape
    if not tattn_shape_ok(x, heads, N, F, HW):
        raise H.MMDError(f"tattn_block: unsupported launch (x {tuple(x.shape)} {x.dtype}, heads={heads} N={N} F={F} HW={HW})")
    out = alloc(M, C, dtype=x.dtype, device=x.device) if out is None else out
    _chk2d(out)
    if out.data_ptr() == x.data_ptr():
        raise H.MMDError("tattn_block: in-place is not supported")
    att = bpre = mid = None
    if pre is not None:
        att, bpre, mid = pre
        _chk2d(att), _chk2d(mid)
        if (att.shape != x.shape or mid.shape != x.shape or att.dtype != x.dtype or mid.dtype != x.dtype
                or len({t.data_ptr() for t in (x, att, mid, out)}) != 4 or wf.numel() * 2 != H.lib().mmd_tattn_weight_bytes(C, 1)):
            raise H.MMDError("tattn_block: the front stage needs att / mid of x's shape, four distinct buffers and weights packed with wpre")
    elif wf.numel() * 2 != H.lib().mmd_tattn_weight_bytes(C, 0):
        raise H.MMDError("tattn_block: weights packed with a front stage need pre=(att, bias_pre, mid)")
    sp, sld = (None, 0) if stats is None else _stats_args(stats, M, C)
    npre = 0 if pre is None else 1
    flops = 2 * M * C * (4 + npre) * C + 4 * M * F * C
    nbytes = 2 * ((3 + 2 * npre) * M * C) + 2 * (4 + npre) * C * C
    _dispatch("mmd_tattn_block", x.data_ptr(), x.stride(0), H.ptr(att), 0 if att is None else att.stride(0), H.ptr(mid),
              0 if mid is None else mid.stride(0), wf.data_ptr(), H.ptr(bpre), bias_qkv.data_ptr(), bias_proj.data_ptr(), gamma.data_ptr(),
              beta.data_ptr(), GN_EPS, out.data_ptr(), out.stride(0), N, F, HW, C, heads, sp, sld,
              meta=(f"tattn_block<bf16{',pre' if npre else ''}>[M={M},C={C}]", flops, nbytes))
    return out


def attn(q, kv, out, heads, ch, nb, G, q_rows_per_batch, q_per_group, k_rows_per_batch, k_per_group, win,
         q_off=0, k_off=None, v_off=None, shift_dev=None, impl=0):
    """See include/mmd.h: mmd_attn_fwd.  q/kv are qkv GEMM outputs [rows, 3C]; out [q rows, C]."""
    _chk2d(q), _chk2d(kv), _chk2d(out)
    C = heads * ch
    k_off = C if k_off is None else k_off
    v_off = 2 * C if v_off is None else v_off
    _dispatch("mmd_attn_fwd", H.dt_of(q), q.data_ptr(), q.stride(0), q_off, kv.data_ptr(), kv.stride(0), k_off, v_off,
           out.data_ptr(), out.stride(0), heads, ch, nb, G, q_rows_per_batch, q_per_group, k_rows_per_batch, k_per_group,
           win, H.ptr(shift_dev), impl,
           meta=(f"attn_fwd[ch={ch},q={q_per_group},k={win * k_per_group},G={nb * G},h={heads}]", 4 * nb * q_rows_per_batch * win * k_per_group * C,
                 q.element_size() * nb * (2 * q_rows_per_batch * C + 2 * G * win * k_per_group * C)))
    return out


def attn_small(qkv, out, C, heads, geom: Geom):
    _chk2d(qkv), _chk2d(out)
    _dispatch("mmd_attn_small_fwd", H.dt_of(qkv), qkv.data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0), C, heads,
           *geom.args(), meta=("attn_small", 4 * geom.S * geom.Tn * geom.Tn * C, 4 * geom.S * geom.Tn * C * qkv.element_size()))
    return out


def attn_small_bwd(qkv, dout, dqkv, C, heads, geom: Geom):
    _chk2d(qkv), _chk2d(dout), _chk2d(dqkv)
    _dispatch("mmd_attn_small_bwd", H.dt_of(qkv), qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0), dqkv.data_ptr(),
              dqkv.stride(0), C, heads, *geom.args(),
              meta=("attn_small_bwd", 10 * geom.S * geom.Tn * geom.Tn * C, 8 * geom.S * geom.Tn * C * qkv.element_size()))
    return dqkv


def resample(x, out, NF, Hh, Ww, fh, fw, mode, scale=1.0, stats=None):
    """mode 0 avg-pool / 1 nearest-upsample by (1, fh, fw); Hh, Ww describe the input rows (nf, h, w).  stats (bf16, scale 1): the
    record view [out rows / 64, C / 4, 2] that receives the GroupNorm statistics of the output (include/mmd.h: mmd_resample_stats)."""
    _chk2d(x), _chk2d(out)
    nbytes = (x.shape[0] + out.shape[0]) * x.shape[1] * x.element_size()
    if stats is not None:
        if x.dtype != torch.bfloat16 or out.dtype != torch.bfloat16 or float(scale) != 1.0:
            raise H.MMDError("resample: output statistics need bf16 rows and scale 1")
        sp, sld = _stats_args(stats, out.shape[0], x.shape[1])
        _dispatch("mmd_resample_stats", x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[1], NF, Hh, Ww, fh, fw, mode, sp, sld,
                  meta=("resample", 0, nbytes))
        return out
    _dispatch("mmd_resample", H.dt_of(x), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[1], NF, Hh, Ww,
           fh, fw, mode, float(scale), meta=("resample", 0, nbytes))
    return out


def copy2d(x, out):
    _chk2d(x), _chk2d(out)
    es = x.element_size()
    _dispatch("mmd_copy2d", x.data_ptr(), x.stride(0) * es, out.data_ptr(), out.stride(0) * es, x.shape[0], x.shape[1] * es)
    return out


def temb(t, dim, W0, b0, W2, b2, out_silu, out_raw=None):
    kind = {torch.int64: 0, torch.int32: 1, torch.float32: 2}.get(t.dtype)
    if kind is None:
        raise H.MMDError(f"timesteps must be int64/int32/float32, got {t.dtype}")
    H.require_cuda(t, W0)
    _dispatch("mmd_temb_fwd", t.data_ptr(), kind, t.shape[0], dim, W0.data_ptr(), b0.data_ptr(), W2.data_ptr(), b2.data_ptr(),
           out_silu.data_ptr(), H.ptr(out_raw))
    return out_silu


def linear(x, W, b, out):
    _dispatch("mmd_linear_fwd", x.data_ptr(), W.data_ptr(), H.ptr(b), out.data_ptr(), x.shape[0], x.shape[1], W.shape[0])
    return out


def stem_conv(x, w, bias, out, N, F, Cin, Hh, Ww, taps):
    """x fp32 contiguous [N,F,Cin,H,W]; w fp32 [ntaps, Cin, Cout]; out rows [N*F*H*W, Cout]."""
    _chk2d(out)
    arr, nt = H.taps_array(taps)
    _dispatch("mmd_stem_conv", H.dt_of(out), x.data_ptr(), w.data_ptr(), H.ptr(bias), out.data_ptr(), out.stride(0), N, F, Cin,
           Hh, Ww, out.shape[1], nt, arr, meta=("stem_conv", 2 * out.shape[0] * out.shape[1] * Cin * nt,
                                                 x.numel() * 4 + out.shape[0] * out.shape[1] * out.element_size()))
    return out


def head_conv(x, w, bias, out, N, F, Hh, Ww, taps):
    """x rows [N*F*H*W, Cin]; w fp32 [ntaps, Cin, Co]; out fp32 contiguous [N,F,Co,H,W]."""
    _chk2d(x)
    arr, nt = H.taps_array(taps)
    _dispatch("mmd_head_conv", H.dt_of(x), x.data_ptr(), x.stride(0), w.data_ptr(), H.ptr(bias), out.data_ptr(), N, F,
           x.shape[1], Hh, Ww, w.shape[2], nt, arr,
           meta=("head_conv", 2 * x.shape[0] * x.shape[1] * w.shape[2] * nt, x.shape[0] * x.shape[1] * x.element_size() + out.numel() * 4))
    return out


# The head for few output channels as GEMM + gather (include/mmd.h: mmd_head_gemm / mmd_head_gather): GroupNorm + SiLU in the GEMM's
# operand registers, per-row products on the matrix cores, then a coalesced gather over the taps.  MMD_HEAD_GEMM=0: gn_apply + the direct
# kernel (A/B).
_HEAD_GEMM = os.environ.get("MMD_HEAD_GEMM", "1") != "0"


def head_gemm_ok(x, w, geom: Geom):
    """Launches the GEMM + gather head accepts: bf16 rows of 128 channels, ntaps * Co <= 96, Co in (1, 2, 3, 4, 6), contiguous slices that
    are multiples of 128 rows."""
    return (_HEAD_GEMM and x.dtype == torch.bfloat16 and x.shape[1] == 128 and w.shape[0] * w.shape[2] <= 96 and w.shape[2] in (1, 2, 3, 4, 6)
            and geom.inner == 1 and geom.tstride == 1 and geom.outer_stride == geom.Tn and geom.Tn % 128 == 0 and geom.S * geom.Tn == x.shape[0]
            and x.stride(0) % 8 == 0)


def head_gemm_pack(w):
    """w fp32 [ntaps, Cin, Co] (pack_edge_weight) -> the (hi, lo) bf16 weight image of mmd_head_gemm: [2][3][Cin / 16][64 lanes][8],
    lane (l31, half) of (block ob, k-step cg) = W[32 ob + l31][16 cg + 8 half .. + 8] with W[tap Co + co][ci] = w[tap][ci][co]."""
    H.require_cuda(w)
    nt, Cin, Co = w.shape
    if nt * Co > 96 or Cin != 128:
        raise H.MMDError(f"head_gemm_pack: needs ntaps * Co <= 96 and Cin == 128, got {tuple(w.shape)}")
    full = torch.zeros(96, Cin, dtype=torch.float32, device=w.device)
    full[: nt * Co] = w.permute(0, 2, 1).reshape(nt * Co, Cin)
    hi = full.to(torch.bfloat16)
    lo = (full - hi.float()).to(torch.bfloat16)
    img = torch.stack([hi, lo]).view(2, 3, 32, Cin // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()    # (hl, ob, cg, half, l31, e)
    assert img.numel() * 2 == H.lib().mmd_head_gemm_weight_bytes(Cin)
    return img.view(-1)


def head_gemm(x, a, b, geom: Geom, act, wimg, P, NO):
    """P[o, m] = sum_ci W[o, ci] act(x[m, ci] a + b) (include/mmd.h: mmd_head_gemm); P fp32 [NO, M]."""
    _chk2d(x)
    M, Cin = x.shape
    if P.dtype != torch.float32 or P.numel() < NO * M or not P.is_contiguous():
        raise H.MMDError("head_gemm: the workspace must be a contiguous fp32 buffer of ntaps * Co * M elements")
    if wimg.numel() * 2 != H.lib().mmd_head_gemm_weight_bytes(Cin):
        raise H.MMDError("head_gemm: the weight image does not match Cin")
    _dispatch("mmd_head_gemm", x.data_ptr(), x.stride(0), M, Cin, a.data_ptr(), b.data_ptr(), geom.S, geom.Tn, 1 if act else 0, wimg.data_ptr(),
              P.data_ptr(), NO, meta=(f"head_gemm[M={M},K={Cin},N={NO}]", 4 * M * Cin * 96, 2 * M * Cin + 4 * NO * M))
    return P


def head_gather(P, bias, out, N, F, Hh, Ww, Co, taps):
    """out fp32 [N, F, Co, H, W] = bias + the tap sum of the product planes P [ntaps * Co, N F H W] (include/mmd.h: mmd_head_gather)."""
    arr, nt = H.taps_array(taps)
    _dispatch("mmd_head_gather", P.data_ptr(), H.ptr(bias), out.data_ptr(), N, F, Hh, Ww, Co, nt, arr,
              meta=("head_gather", 0, 4 * nt * Co * N * F * Hh * Ww + out.numel() * 4))
    return out


def ddpm_update(x, model_out, noise, out, tables, t, F, C, HW, flags, x0_out=None, mean_out=None, logvar_out=None):
    H.require_cuda(x, model_out, noise, out, tables, t)
    _dispatch("mmd_ddpm_update", x.data_ptr(), model_out.data_ptr(), H.ptr(noise), H.ptr(out), H.ptr(x0_out),
              H.ptr(mean_out), H.ptr(logvar_out), tables.data_ptr(), t.data_ptr(), tables.shape[1], x.shape[0], F, C, HW, flags,
              meta=("ddpm_update", 0, 16 * x.numel()))
    return out


def ddim_update(x, model_out, noise, out, tables, tab3, t, F, C, HW, flags, eta, x0_out=None):
    H.require_cuda(x, model_out, tables, tab3, t)
    _dispatch("mmd_ddim_update", x.data_ptr(), model_out.data_ptr(), H.ptr(noise), H.ptr(out), H.ptr(x0_out), tables.data_ptr(),
              tab3.data_ptr(), t.data_ptr(), tables.shape[1], x.shape[0], F, C, HW, flags, float(eta),
              meta=("ddim_update", 0, 16 * x.numel()))
    return out


def lincomb_t(a, b, out, ca, cb, cs, t):
    """out[n] = (ca[t_n] a[n] + cb[t_n] b[n]) cs[t_n] (None table = 1, b may be None); fp32 contiguous tensors."""
    H.require_cuda(a, out, t)
    _dispatch("mmd_lincomb_t", a.data_ptr(), H.ptr(b), out.data_ptr(), H.ptr(ca), H.ptr(cb), H.ptr(cs), t.data_ptr(), a.shape[0],
              a[0].numel(), meta=("lincomb_t", 0, 12 * a.numel()))
    return out


def lincomb(a, ca, b=None, cb=0.0, c=None, cc=0.0, out=None):
    """out = ca a + cb b + cc c with host scalars; fp32 contiguous tensors of one size."""
    H.require_cuda(a)
    out = torch.empty_like(a) if out is None else out
    _dispatch("mmd_lincomb", a.data_ptr(), float(ca), H.ptr(b), float(cb), H.ptr(c), float(cc), out.data_ptr(), a.numel(),
              meta=("lincomb", 0, 16 * a.numel()))
    return out


def bilinear_concat(x, low, out):
    """out [N,2C,H,W] = concat(x, bilinear-upsampled low) (fp32 API layout): the SR model's input."""
    H.require_cuda(x, low, out)
    N, C, Hh, Ww = x.shape
    _dispatch("mmd_bilinear_concat", x.data_ptr(), low.data_ptr(), out.data_ptr(), N, C, Hh, Ww, low.shape[2], low.shape[3],
              meta=("bilinear_concat", 0, 4 * (x.numel() + low.numel() + out.numel())))
    return out


def bilinear_concat_rows(x, low, out):
    """out [N*H*W, Cpad] (bf16 / fp32 rows) = [x | bilinear(low) | 0]: the SR model's input in the layout of the implicit-GEMM stem."""
    H.require_cuda(x, low, out)
    N, C, Hh, Ww = x.shape
    _chk2d(out)
    if out.shape[0] != N * Hh * Ww or not out.is_contiguous():
        raise H.MMDError("bilinear_concat_rows: out must be contiguous [N*H*W, Cpad]")
    _dispatch("mmd_bilinear_concat_rows", H.dt_of(out), x.data_ptr(), low.data_ptr(), out.data_ptr(), N, C, Hh, Ww, low.shape[2], low.shape[3],
              out.shape[1], meta=("bilinear_concat_rows", 0, 4 * (x.numel() + low.numel()) + out.numel() * out.element_size()))
    return out


def abs_quantile(x, q):
    """per-sample q-quantile of |x| (fp32 contiguous [N, ...]) -> fp32 [N]."""
    H.require_cuda(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _dispatch("mmd_abs_quantile", x.data_ptr(), x.shape[0], x[0].numel(), float(q), out.data_ptr(), meta=("abs_quantile", 0, 20 * x.numel()))
    return out


def clamp_scale_(x, s, max_val):
    H.require_cuda(x, s)
    _dispatch("mmd_clamp_scale", x.data_ptr(), s.data_ptr(), float(max_val), x.shape[0], x[0].numel(), meta=("clamp_scale", 0, 8 * x.numel()))
    return x


def dpm_err(hi, lo, prev, atol, rtol, out):
    """out[n] (fp64, pre-zeroed) += squared scaled difference of two solver orders (adaptive step-size control)."""
    H.require_cuda(hi, lo, prev, out)
    _dispatch("mmd_dpm_err", hi.data_ptr(), lo.data_ptr(), prev.data_ptr(), float(atol), float(rtol), hi.shape[0], hi[0].numel(),
              out.data_ptr(), meta=("dpm_err", 0, 12 * hi.numel()))
    return out


def ddpm_update_bwd(x, model_out, dsample, dx, dmo, tables, t, flags):
    H.require_cuda(x, model_out, dsample, tables, t)
    _dispatch("mmd_ddpm_update_bwd", x.data_ptr(), model_out.data_ptr(), dsample.data_ptr(), H.ptr(dx), H.ptr(dmo), tables.data_ptr(),
              t.data_ptr(), tables.shape[1], x.shape[0], x[0].numel(), flags, meta=("ddpm_update_bwd", 0, 20 * x.numel()))


def q_sample(x0, eps, out, tab2, t):
    H.require_cuda(x0, eps, out, tab2, t)
    _dispatch("mmd_q_sample", x0.data_ptr(), eps.data_ptr(), out.data_ptr(), tab2.data_ptr(), t.data_ptr(), tab2.shape[1],
           x0.shape[0], x0[0].numel())
    return out


def loss_terms(model_out, target, tables, t, F, C, HW, flags, x0=None, xt=None, vb_scale=1.0):
    """Per-sample (mse, vb) of one stream on API-layout fp32 tensors; vb is None without the learned-range flag (4)."""
    H.require_cuda(model_out, target, tables, t)
    N = model_out.shape[0]
    mse = torch.empty(N, dtype=torch.float32, device=model_out.device)
    vb = torch.empty(N, dtype=torch.float32, device=model_out.device) if flags & 4 else None
    ws = torch.empty(H.lib().mmd_loss_workspace_bytes(N) // 8, dtype=torch.float64, device=model_out.device)
    _dispatch("mmd_loss_terms", H.ptr(x0), H.ptr(xt), model_out.data_ptr(), target.data_ptr(), tables.data_ptr(), t.data_ptr(),
              tables.shape[1], N, F, C, HW, flags, float(vb_scale), mse.data_ptr(), H.ptr(vb), ws.data_ptr(),
              meta=("loss_terms", 0, 8 * target.numel()))
    return mse, vb


# ------------------------------------------------------------------ training step (backward) wrappers
def loss_terms_bwd(model_out, target, tables, t, F, C, HW, flags, dmse, dvb, g, x0=None, xt=None, vb_scale=1.0):
    """g (like model_out) = d(sum dmse*mse + dvb*vb)/d model_out of loss_terms; see include/mmd.h."""
    H.require_cuda(model_out, target, tables, t, dmse, g)
    _dispatch("mmd_loss_terms_bwd", H.ptr(x0), H.ptr(xt), model_out.data_ptr(), target.data_ptr(), tables.data_ptr(), t.data_ptr(),
              tables.shape[1], model_out.shape[0], F, C, HW, flags, float(vb_scale), dmse.data_ptr(), H.ptr(dvb), g.data_ptr(),
              meta=("loss_terms_bwd", 0, 20 * target.numel()))
    return g


def conv_wgrad(dy, x, dW, db, taps, dims, torch_layout=False):
    """dW fp32 += dy^T gather(x) in [Cout, ntaps*Cin] (packed) or, with torch_layout, in the parameter's own [Cout, Cin, *k]
    layout (so dW may be the parameter's .grad); db fp32 [Cout] += colsum(dy).  Accumulating: the caller owns the zeroing."""
    _chk2d(dy), _chk2d(x)
    arr, nt = H.taps_array(taps)
    _dispatch("mmd_conv_wgrad", H.dt_of(x), dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dW.data_ptr(), H.ptr(db),
              dy.shape[0], dy.shape[1], x.shape[1], nt, arr, int(dims[0]), int(dims[1]), int(dims[2]), 1 if torch_layout else 0,
              meta=("conv_wgrad", 2 * dy.shape[0] * dy.shape[1] * x.shape[1] * nt, 0))


# GroupNorm backward workspaces kept zero between calls (mmd_gn_bwd_ws0): one per (device, launch stream, size); MMD_GN_BWD_WS0=0 = the
# per-call workspace with its fill launch (A/B)
_GN_BWD_WS0 = os.environ.get("MMD_GN_BWD_WS0", "1") != "0"
_gn_bwd_ws = {}


def gn_bwd(x, dy, dx, geom: Geom, a, b, mr, gamma, beta, film, act, dgamma, dbeta, dfilm):
    _chk2d(x), _chk2d(dy), _chk2d(dx)
    C = x.shape[1]
    n = geom.S * C * 2 + geom.S * 64
    if _GN_BWD_WS0 and _recorder is None:
        key = (str(x.device), H.stream_handle(), n)
        ws = _gn_bwd_ws.get(key)
        if ws is None:
            ws = _gn_bwd_ws[key] = torch.zeros(n, dtype=torch.float32, device=x.device)
        entry = "mmd_gn_bwd_ws0"
    else:
        ws = torch.empty(n, dtype=torch.float32, device=x.device)
        entry = "mmd_gn_bwd"
    _dispatch(entry, H.dt_of(x), x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), dx.data_ptr(), dx.stride(0),
              x.shape[0], C, *geom.args(), a.data_ptr(), b.data_ptr(), mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              H.ptr(film), 0 if film is None else film.stride(0), 1 if act else 0, dgamma.data_ptr(), dbeta.data_ptr(),
              H.ptr(dfilm), 0 if dfilm is None else dfilm.stride(0), ws.data_ptr(), meta=("gn_bwd", 0, 3 * x.numel() * x.element_size()))


def attn_bwd(q, q_off, kv, k_off, v_off, o, do, dq, dq_off, dkv, dk_off, dv_off, heads, ch, nb, G, qgeo, q_total, q_per_group,
             kgeo, k_mod, k_per_group, win, shift_dev):
    """See include/mmd.h: mmd_attn_bwd.  qgeo / kgeo = (inner, outer_stride, inner_stride, tstride) of the unit bases."""
    for t in (q, kv, o, do, dq, dkv):
        _chk2d(t)
    lse = torch.empty(q.shape[0] * heads, dtype=torch.float32, device=q.device)
    dsum = torch.empty(q.shape[0] * heads, dtype=torch.float32, device=q.device)
    _dispatch("mmd_attn_bwd", H.dt_of(q), q.data_ptr(), q.stride(0), q_off, kv.data_ptr(), kv.stride(0), k_off, v_off,
              o.data_ptr(), o.stride(0), do.data_ptr(), do.stride(0), dq.data_ptr(), dq.stride(0), dq_off, dkv.data_ptr(),
              dkv.stride(0), dk_off, dv_off, lse.data_ptr(), dsum.data_ptr(), heads, ch, nb, G, *[int(v) for v in qgeo], q_total,
              q_per_group, *[int(v) for v in kgeo], k_mod, k_per_group, win, H.ptr(shift_dev),
              meta=("attn_bwd", 10 * nb * q_total * win * k_per_group * heads * ch, 0))


MFMA_HEADS = (16, 32, 48, 64, 96, 128)


def attn_mfma_ok(t, ch):
    return t.dtype == torch.bfloat16 and ch in MFMA_HEADS and t.stride(0) % 8 == 0


def attn_lse(q, kv, out, lse, heads, ch, nb, G, q_rows_per_batch, q_per_group, k_rows_per_batch, k_per_group, win, shift_dev=None):
    """bf16 MFMA forward that also returns the per-(row, head) log2-domain log-sum-exp (training forward)."""
    _chk2d(q), _chk2d(kv), _chk2d(out)
    C = heads * ch
    _dispatch("mmd_attn_fwd_lse", H.dt_of(q), q.data_ptr(), q.stride(0), 0, kv.data_ptr(), kv.stride(0), C, 2 * C, out.data_ptr(),
              out.stride(0), heads, ch, nb, G, q_rows_per_batch, q_per_group, k_rows_per_batch, k_per_group, win, H.ptr(shift_dev),
              lse.data_ptr(), meta=("attn_fwd_lse", 4 * nb * q_rows_per_batch * win * k_per_group * C, 0))
    return out


def attn_bwd_mfma(q, kv, o, do, dq, dq_off, dkv, dk_off, dv_off, lse, heads, ch, nb, G, q_rows_per_batch, q_per_group,
                  k_rows_per_batch, k_per_group, win, shift_dev=None):
    for t in (q, kv, o, do, dq, dkv):
        _chk2d(t)
    C = heads * ch
    dsum = torch.empty(q.shape[0] * heads, dtype=torch.float32, device=q.device)
    _dispatch("mmd_attn_bwd_mfma", q.data_ptr(), q.stride(0), 0, kv.data_ptr(), kv.stride(0), C, 2 * C, o.data_ptr(), o.stride(0),
              do.data_ptr(), do.stride(0), dq.data_ptr(), dq.stride(0), dq_off, dkv.data_ptr(), dkv.stride(0), dk_off, dv_off,
              lse.data_ptr(), dsum.data_ptr(), heads, ch, nb, G, q_rows_per_batch, q_per_group, k_rows_per_batch, k_per_group, win,
              H.ptr(shift_dev), meta=("attn_bwd_mfma", 10 * nb * q_rows_per_batch * win * k_per_group * C, 0))


def silu(x, dy, out):
    _dispatch("mmd_silu", H.dt_of(x), x.data_ptr(), H.ptr(dy), out.data_ptr(), x.numel(), meta=("silu", 0, 2 * x.numel() * x.element_size()))
    return out


def dropout(x, mask, scale, out):
    _dispatch("mmd_dropout", H.dt_of(x), x.data_ptr(), mask.data_ptr(), float(scale), out.data_ptr(), x.numel(),
              meta=("dropout", 0, 2 * x.numel() * x.element_size()))
    return out


def mse_grad(out, target, w, g):
    _dispatch("mmd_mse_grad", out.data_ptr(), target.data_ptr(), w.data_ptr(), g.data_ptr(), out.shape[0], out[0].numel(),
              meta=("mse_grad", 0, 12 * out.numel()))
    return g


def adamw_step(p, g, m, v, ema, lr, beta1, beta2, eps, weight_decay, step, ema_rate=0.0):
    _dispatch("mmd_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), H.ptr(ema), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), int(step), float(ema_rate), meta=("adamw", 0, 28 * p.numel()))


def timestep_embedding(t, dim, out):
    kind = {torch.int64: 0, torch.int32: 1, torch.float32: 2}.get(t.dtype)
    if kind is None:
        raise H.MMDError(f"timesteps must be int64/int32/float32, got {t.dtype}")
    _dispatch("mmd_timestep_embedding", t.data_ptr(), kind, t.shape[0], dim, out.data_ptr())
    return out


def pack_conv_weight(w: torch.Tensor, dtype) -> torch.Tensor:
    """[Cout, Cin, *k] (torch conv layout) -> [Cout, ntaps*Cin] with K index = tap*Cin + ci (tap = row-major k)."""
    Cout, Cin = w.shape[0], w.shape[1]
    return w.reshape(Cout, Cin, -1).permute(0, 2, 1).reshape(Cout, -1).to(dtype).contiguous()


def pack_edge_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, *k] -> fp32 [ntaps, Cin, Cout] for the stem / head kernels."""
    Cout, Cin = w.shape[0], w.shape[1]
    return w.reshape(Cout, Cin, -1).permute(2, 1, 0).float().contiguous()


def cast(x, out, scale=1.0):
    """out = (out.dtype)(x * scale), fp32 <-> bf16 flat buffers (include/mmd.h: mmd_cast)."""
    H.require_cuda(x, out)
    if x.numel() != out.numel() or not x.is_contiguous() or not out.is_contiguous():
        raise H.MMDError("cast: contiguous buffers of equal length expected")
    _dispatch("mmd_cast", x.data_ptr(), H.dt_of(x), out.data_ptr(), H.dt_of(out), float(scale), x.numel())
    return out
