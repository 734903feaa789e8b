#!/usr/bin/env python3
"""3x3 convs of the ds1 / ds2 levels (Landscape model, batch 4, bf16) and of the SR model: direct-to-LDS tile 129, halo tiles 130
(8 x 16 patches, two 4-wave blocks per CU) and 133 (16 x 16 patches, one 8-wave block per CU, three-slot weight ring), each plain and
with the input GroupNorm + SiLU fused (gn) against gn_apply + conv."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

SHAPES = [  # name, frames, H, W, Cin, Cout
    ("ds1 128->128", 64, 64, 64, 128, 128),
    ("ds1 256->128", 64, 64, 64, 256, 128),
    ("ds1 384->128", 64, 64, 64, 384, 128),
    ("ds2 128->256", 64, 32, 32, 128, 256),
    ("ds2 256->256", 64, 32, 32, 256, 256),
    ("ds2 640->256", 64, 32, 32, 640, 256),
    ("sr 256x256 192->192", 16, 256, 256, 192, 192),
]


def timed(fn, n=10):
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    for _ in range(2):
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
    dt = torch.bfloat16
    for name, D0, Hh, Ww, Cin, Cout in SHAPES:
        M = D0 * Hh * Ww
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
        w = (torch.randn(Cout, Cin * 9, device="cuda", generator=g) * (Cin * 9) ** -0.5).to(dt)
        b = torch.randn(Cout, device="cuda", generator=g)
        gamma, beta = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        N = 4 if D0 % 4 == 0 else 1
        geom = ops.Geom.per_sample(N, M // N)
        ga, gb = ops.gn_stats(x, gamma, beta, geom)
        dims = (D0, Hh, Ww)
        flops = 2.0 * M * Cout * Cin * 9
        y = torch.empty(M, Cout, device="cuda", dtype=dt)
        xn = torch.empty_like(x)
        line = f"{name:22s} M={M:7d} K={Cin*9:5d} N={Cout:4d}"
        ref = None
        for tile in (129, 130, 133):
            us = timed(lambda: ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=dims, out=y, tile=tile))
            if tile == 130:
                ref = y.clone()
            same = "" if tile != 133 else (" =" if torch.equal(y, ref) else " DIFF")
            line += f" | t{tile}: {us:6.1f}us {flops/us/1e6:5.0f}TF{same}"
        us_ap = timed(lambda: ops.gn_apply(x, ga, gb, geom, act=True, out=xn))
        line += f" | apply {us_ap:5.1f}us"
        for tile in (130, 133):
            us = timed(lambda: ops.gn_conv_gemm(x, ga, gb, geom, True, w, b, ops.TAPS_SPATIAL, dims, out=y, tile=tile))
            line += f" | gn{tile}: {us:6.1f}us {flops/us/1e6:5.0f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
