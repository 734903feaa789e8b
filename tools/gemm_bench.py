#!/usr/bin/env python3
"""Micro-benchmark of mmd_conv_gemm variants (tile 64 / 128 register-staged / 129 direct-to-LDS) on the shapes the
Landscape model actually launches at batch 4.  Checks each variant against the 64-tile result first."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

SHAPES = [  # name, M, Cin, taps, dims, Cout, residual
    ("3x3 ds1 128->128", 262144, 128, ops.TAPS_SPATIAL, (64, 64, 64), 128, False),
    ("3x3 ds1 256->128", 262144, 256, ops.TAPS_SPATIAL, (64, 64, 64), 128, False),
    ("k3t ds1 128->128", 262144, 128, ops.TAPS_TEMPORAL, (16, 4096, 1), 128, False),
    ("k3t ds1 128->128 (N,F,HW)", 262144, 128, ops.TAPS_TEMPORAL_D1, (4, 16, 4096), 128, False),
    ("k3t ds2 256->256 (N,F,HW)", 65536, 256, ops.TAPS_TEMPORAL_D1, (4, 16, 1024), 256, False),
    ("3x3 ds2 256->256", 65536, 256, ops.TAPS_SPATIAL, (64, 32, 32), 256, False),
    ("3x3 ds4 384->384", 16384, 384, ops.TAPS_SPATIAL, (64, 16, 16), 384, False),
    ("3x3 ds8 512->512", 4096, 512, ops.TAPS_SPATIAL, (64, 8, 8), 512, False),
    ("1x1 ds1 128->128 +res", 262144, 128, ops.TAPS_1, (1, 1, 1), 128, True),
    ("qkv ds2 256->768", 65536, 256, ops.TAPS_1, (1, 1, 1), 768, False),
    ("proj ds2 256->256 +res", 65536, 256, ops.TAPS_1, (1, 1, 1), 256, True),
    ("qkv ds4 384->1152", 16384, 384, ops.TAPS_1, (1, 1, 1), 1152, False),
    ("1x1 ds8 512->512", 4096, 512, ops.TAPS_1, (1, 1, 1), 512, True),
    ("audio k3 d4 128->128", 102400, 128, ops.taps_audio(4), (25600, 1, 1), 128, False),
    ("proj ds2 256->256", 65536, 256, ops.TAPS_1, (1, 1, 1), 256, False),
    ("proj ds4 384->384 +res", 16384, 384, ops.TAPS_1, (1, 1, 1), 384, True),
    ("skip ds1 384->128", 262144, 384, ops.TAPS_1, (1, 1, 1), 128, False),
    ("skip ds2 768->256", 65536, 768, ops.TAPS_1, (1, 1, 1), 256, False),
    ("qkv ds8 512->1536", 4096, 512, ops.TAPS_1, (1, 1, 1), 1536, False),
    ("audio qkv 256->768", 25600, 256, ops.TAPS_1, (1, 1, 1), 768, False),
]
TILES = (64, 129, 130, 131, 132, 133)     # 130 / 133 = halo tiles (spatial 3x3 only), 131 = row strip (K <= 512), 132 = four-slot ring


def main():
    dt = torch.bfloat16
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    for name, M, Cin, taps, dims, Cout, res in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
        w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(dt)
        b = torch.randn(Cout, device="cuda", generator=g)
        r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
        flops = 2.0 * M * Cout * Cin * len(taps)
        nbytes = 2 * (M * Cin + M * Cout * (2 if res else 1) + Cout * Cin * len(taps))
        ref = None
        line = f"{name:26s} M={M:6d} K={Cin*len(taps):5d} N={Cout:4d}"
        for tile in TILES:
            if tile in (130, 133) and not (len(taps) == 9 and dims[1] % 16 == 0 and dims[2] % 16 == 0 and Cin % 64 == 0):
                continue
            if tile == 131 and not ops.strip_tile_ok(x, Cout, taps):
                continue
            y = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile)
            if ref is None:
                ref = y.clone()
            err = float((y.float() - ref.float()).abs().max())
            for _ in range(2):
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
            H.call("mmd_event_record", ev[0], st)
            n = 10
            for _ in range(n):
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
            H.call("mmd_event_record", ev[1], st)
            ms = ctypes.c_float()
            H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
            us = ms.value / n * 1000
            line += f" | t{tile}: {us:7.1f}us {flops/us/1e6:6.0f}TF {nbytes/us/1e3:5.0f}GB/s e={err:.0e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
