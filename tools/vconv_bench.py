#!/usr/bin/env python3
"""mmd_vconv2d1d (fused VideoConv 2d+1d with the in_layers norm in LDS and output statistics) against the launches it replaces -
gn_apply-in-the-halo 3x3 conv (mmd_gn_conv_gemm, tile 130 / 133) + temporal k=3 on the strip with statistics - on the ds1 layer
shapes of the Landscape model at batch 4.  HIP-event times, 20 launches each, interleaved rounds."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402


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
    return ms.value / n * 1000


def main():
    N, F, Hh = 4, 16, 64
    M = N * F * Hh * Hh
    for Cin in (128, 256, 384):
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
        ws = (torch.randn(128, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(torch.bfloat16)
        wt = (torch.randn(128, 384, device="cuda", generator=g) * 384 ** -0.5).to(torch.bfloat16)
        bs, bt = torch.randn(128, device="cuda", generator=g), torch.randn(128, device="cuda", generator=g)
        a, b = torch.rand(N, Cin, device="cuda", generator=g) + 0.5, torch.randn(N, Cin, device="cuda", generator=g) * 0.5
        geom = ops.Geom.per_sample(N, F * Hh * Hh)
        wf = ops.vconv_pack(ws, wt)
        rec = torch.zeros(M // 64, 32, 2, device="cuda")
        t1, y0, y1 = (torch.empty(M, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        flops = 2.0 * M * 128 * (9 * Cin + 384)

        def two():
            ops.gn_conv_gemm(x, a, b, geom, True, ws, bs, ops.TAPS_SPATIAL, (N * F, Hh, Hh), out=t1)
            ops.conv_gemm(t1, wt, bt, taps=ops.TAPS_TEMPORAL, dims=(F, Hh * Hh, 1), out=y0, stats=rec)

        def fused():
            ops.vconv2d1d(x, wf, bs, bt, N, F, Hh, Hh, a=a, b=b, geom=geom, act=True, out=y1, stats=rec)

        def fused_plain():
            ops.vconv2d1d(x, wf, bs, bt, N, F, Hh, Hh, out=y1)

        two(); fused()
        err = float((y1.float() - y0.float()).norm() / y0.float().norm())
        r = {"two-launch": [], "fused": [], "fused (no norm, no stats)": []}
        for _ in range(3):
            r["two-launch"].append(timed(two))
            r["fused"].append(timed(fused))
            r["fused (no norm, no stats)"].append(timed(fused_plain))
        line = f"vconv ds1 {Cin}->128 (M={M}, {flops / 1e9:.1f} GFLOP) rel-L2 fused vs two-launch {err:.1e}"
        for k, v in r.items():
            line += f" | {k}: {min(v):6.1f} us {flops / min(v) / 1e6:5.0f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
