#!/usr/bin/env python3
"""Row-strip 1x1 GEMM (conv_gemm tile 131) against the paths it replaces, on the 1x1-conv shapes of the Landscape model at
batch 4 (bf16): time per launch (HIP events, 20 launches) and equality of the outputs.

  plain    : conv_gemm tile 129 (or 128 with statistics)            vs tile 131
  gn       : gn_conv1x1 tile 128 where the tiled loader applies,
             else gn_apply + conv_gemm tile 129 (two launches)       vs gn_conv1x1 tile 131 (one launch)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

BF = torch.bfloat16
# name, M, Cin, Cout, GroupNorm slices (S, act) or None, residual, statistics
SHAPES = [
    ("v ds2 qkv (spatial attn)", 65536, 256, 768, (64, False), False, False),
    ("v ds2 qkv (cross attn)", 65536, 256, 768, (4, False), False, False),
    ("v ds4 qkv (spatial attn)", 16384, 384, 1152, (64, False), False, False),
    ("v ds2 proj_out + x", 65536, 256, 256, None, True, True),
    ("v ds4 proj_out + x", 16384, 384, 384, None, True, False),
    ("v ds1 res out conv + skip", 262144, 128, 128, (4, True), True, True),
    ("v ds2 res out conv + skip", 65536, 256, 256, (4, True), True, True),
    ("v ds1 up out conv + skip", 262144, 256, 256, (4, True), True, True),
    ("v ds1 skip conv 256->128", 262144, 256, 128, None, False, False),
    ("a ds1 res out conv + skip", 102400, 128, 128, (4, True), True, True),
    ("a ds2 qkv", 25600, 256, 768, (4, False), False, False),
    ("a ds2 res out conv + skip", 25600, 256, 256, (4, True), True, True),
    ("a ds4 qkv", 6400, 384, 1152, (4, False), False, False),
    ("v ds4 proj_out + x (stats)", 16384, 384, 384, None, True, True),
    ("v ds8 qkv", 4096, 512, 1536, (4, False), False, False),
    ("v ds8 proj_out + x (stats)", 4096, 512, 512, None, True, True),
    ("v ds1 temporal k3 (stats)", 262144, 128, 128, "temporal", False, True),
    ("v ds1 temporal k3", 262144, 128, 128, "temporal", False, False),
    ("a ds1 conv k3 d=4 (stats)", 102400, 128, 128, "audio", False, True),
]


def timed(fn, n=20):
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    for _ in range(3):
        fn()
    H.call("mmd_event_record", ev[0], st)
    for _ in range(n):
        fn()
    H.call("mmd_event_record", ev[1], st)
    ms = ctypes.c_float()
    H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
    for e in ev:
        H.lib().mmd_event_destroy(e)
    return ms.value / n * 1000.0


def main():
    only = os.environ.get("STRIP_PROBE_SHAPES")
    tot_old = tot_new = 0.0
    for si, (name, M, Cin, Cout, gn, res, st) in enumerate(SHAPES):
        if only and str(si) not in only.split(","):
            continue
        g = torch.Generator(device="cuda").manual_seed(si)
        x = (torch.randn(M, Cin, device="cuda", generator=g) * 1.3 + 0.2).to(BF)
        w = (torch.randn(Cout, Cin, device="cuda", generator=g) * Cin ** -0.5).to(BF)
        b = torch.randn(Cout, device="cuda", generator=g)
        r = torch.randn(M, Cout, device="cuda", generator=g).to(BF) if res else None
        y_old = torch.empty(M, Cout, device="cuda", dtype=BF)
        y_new = torch.full((M, Cout), float("nan"), device="cuda", dtype=BF)
        rec_old = torch.zeros(M // 64, Cout // 4, 2, device="cuda") if st else None      # one record per 64 rows and quad of channels
        rec_new = torch.zeros(M // 64, Cout // 4, 2, device="cuda") if st else None
        if gn is None or isinstance(gn, str):
            taps, dims = ops.TAPS_1, (1, 1, 1)
            if gn == "temporal":                             # D = (F, HW, 1), 16 frames
                taps, dims = ops.TAPS_TEMPORAL, (16, M // (4 * 16), 1)
            elif gn == "audio":
                taps, dims = ops.taps_audio(4), (M // 4, 1, 1)
            if len(taps) > 1:
                w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(BF)

            def old():
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=129, out=y_old, stats=rec_old)

            def new():
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=131, out=y_new, stats=rec_new)
            how = "tile 129"                                 # the autotuner's choice among the 128-row family at these sizes
            if st:
                us128 = timed(lambda: ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=128, out=y_old, stats=rec_old))
                how += f" (128: {us128:.1f} us)"
        else:
            S, act = gn
            geom = ops.Geom.per_sample(S, M // S)
            gamma, beta = 1 + 0.1 * torch.randn(Cin, device="cuda", generator=g), torch.randn(Cin, device="cuda", generator=g)
            ga, gb = ops.gn_stats(x, gamma, beta, geom)
            tiled = Cin <= 256 and Cout <= 256
            xn = torch.empty_like(x)
            if tiled:
                def old():
                    ops.gn_conv1x1(x, ga, gb, geom, act, w, b, residual=r, tile=128, out=y_old, stats=rec_old)
                how = "gn_conv1x1 tile 128"
            else:
                def old():
                    ops.gn_apply(x, ga, gb, geom, act=act, out=xn)
                    ops.conv_gemm(xn, w, b, residual=r, tile=129, out=y_old)
                how = "gn_apply + tile 129"

            def new():
                ops.gn_conv1x1(x, ga, gb, geom, act, w, b, residual=r, tile=131, out=y_new, stats=rec_new)
        us_old, us_new = timed(old), timed(new)
        torch.cuda.synchronize()
        same = torch.equal(y_old.view(torch.int16), y_new.view(torch.int16))
        err = float((y_new.float() - y_old.float()).norm() / y_old.float().norm())
        serr = ""
        if st:
            yf = y_new.double().view(M // 64, 64, Cout // 4, 4)
            ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], dim=-1)
            serr = f" stats-err {float((rec_new.double() - ref).abs().max() / ref.abs().max()):.1e}"
        kk = Cin * (3 if isinstance(gn, str) else 1)
        nbytes = 2 * (M * Cin + M * Cout * (2 if res else 1))
        tot_old += us_old
        tot_new += us_new
        print(f"{name:28s} M={M:6d} {Cin:3d}->{Cout:4d} | {how:34s} {us_old:7.1f} us | strip {us_new:7.1f} us "
              f"({nbytes / us_new / 1e6:5.2f} TB/s, {2.0 * M * kk * Cout / us_new / 1e6:5.0f} TF/s) | x{us_old / us_new:4.2f} | "
              f"bitwise={same} rel-L2 {err:.1e}{serr}", flush=True)
    print(f"sum: old {tot_old:.0f} us, strip {tot_new:.0f} us")


if __name__ == "__main__":
    main()
