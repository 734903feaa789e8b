#!/usr/bin/env python3
"""mmd_tconv (temporal k = 3 conv with stationary activations, tap shift = DPP lane shift) against conv_gemm on the temporal taps
(autotuned tile) on the three level shapes of the Landscape model at batch 4.  HIP-event times, 20 launches each, best of 3 rounds."""
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
    F = 16
    for N, HW, C in ((4, 1024, 256), (4, 256, 384), (4, 64, 512), (1, 1024, 256)):
        M = N * F * HW
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, 3 * C, device="cuda", generator=g) * (3 * C) ** -0.5).to(torch.bfloat16)
        b = torch.randn(C, device="cuda", generator=g)
        wf = ops.tconv_pack(w)
        y0, y1 = (torch.empty(M, C, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        rec = torch.zeros(M // 64, C // 4, 2, device="cuda")

        def gemm():
            ops.conv_gemm(x, w, b, taps=ops.TAPS_TEMPORAL, dims=(F, HW, 1), out=y0, stats=rec)

        def tc():
            ops.tconv(x, wf, b, C, N, F, HW, out=y1, stats=rec)

        gemm(); tc()
        same = bool(torch.equal(y0, y1))
        t0 = t1 = 1e9
        for _ in range(3):
            t0 = min(t0, timed(gemm))
            t1 = min(t1, timed(tc))
        flops = 2.0 * M * C * 3 * C
        print(f"N={N} HW={HW:5d} C={C} M={M:6d} | conv_gemm {t0:7.1f} us {flops / t0 / 1e6:5.0f} TF/s | tconv {t1:7.1f} us {flops / t1 / 1e6:5.0f} TF/s "
              f"{2 * M * C * 2 / t1 / 1e3:5.0f} GB/s (x + y) | bitwise equal: {same}")


if __name__ == "__main__":
    main()
