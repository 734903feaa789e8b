#!/usr/bin/env python3
"""Micro-benchmark of mmd_aconv (GroupNorm + SiLU + dilated k = 3 AudioConv in one launch) against the two launches it replaces
(mmd_gn_apply + mmd_conv_gemm) on the audio in_layers shapes of the Landscape model at batch 4 (bf16)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

N = 4
SHAPES = [  # L, Cin, Cout, dilation
    (25600, 128, 128, 2), (25600, 384, 128, 4), (6400, 256, 256, 32), (6400, 640, 256, 32), (1600, 384, 384, 256), (1600, 896, 384, 256),
    (400, 512, 512, 2), (400, 1024, 512, 1)]


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
    return ms.value / n * 1000


def main():
    BF = torch.bfloat16
    for L, Cin, Cout, dil in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(N * L, Cin, device="cuda", generator=g).to(BF)
        w = (torch.randn(Cout, 3 * Cin, device="cuda", generator=g) * 0.03).to(BF)
        bias = torch.zeros(Cout, device="cuda")
        a = torch.ones(N, Cin, device="cuda") + 0.1 * torch.randn(N, Cin, device="cuda", generator=g)
        b = 0.1 * torch.randn(N, Cin, device="cuda", generator=g)
        geom = ops.Geom.per_sample(N, L)
        xn = torch.empty_like(x)
        y0 = torch.empty(N * L, Cout, device="cuda", dtype=BF)
        y1 = torch.empty_like(y0)
        rec = torch.zeros(N * L // 64, Cout // 4, 2, device="cuda") if (N * L) % 64 == 0 and L % 64 == 0 else None

        def two():
            ops.gn_apply(x, a, b, geom, act=True, out=xn)
            ops.conv_gemm(xn, w, bias, taps=ops.taps_audio(dil), dims=(L, 1, 1), out=y0, stats=rec)

        def one():
            ops.aconv(x, a, b, w, bias, N, L, dil, act=True, out=y1, stats=rec)

        t2, t1 = timed(two), timed(one)
        tg = timed(lambda: ops.gn_apply(x, a, b, geom, act=True, out=xn))
        torch.cuda.synchronize()
        eq = bool(torch.equal(y0, y1))
        fl = 2.0 * N * L * 3 * Cin * Cout
        print(f"L={L:6d} {Cin:4d}->{Cout:3d} d={dil:3d} | gn_apply + conv_gemm {t2:6.1f} us (gn_apply {tg:5.1f}) | aconv {t1:6.1f} us {fl / t1 / 1e6:5.0f} TF/s | bitwise {eq}", flush=True)


if __name__ == "__main__":
    main()
