#!/usr/bin/env python3
"""Deep-ring direct-to-LDS GEMM (conv_gemm tile 132: four LDS slots, three K steps of DMA in flight, one block per CU) against the
tiles it competes with on the launches that have FEWER TILES THAN THE CHIP HAS BLOCK SLOTS (the ds8 / ds4 levels and the small audio
levels of the Landscape model at batch 4, bf16).  Equality with tile 64 is checked first (all tiled loops share K order and epilogue)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

SHAPES = [  # name, M, Cin, taps, dims, Cout, residual
    ("3x3 ds8 512->512", 4096, 512, ops.TAPS_SPATIAL, (64, 8, 8), 512, False),
    ("3x3 ds8 1024->512", 4096, 1024, ops.TAPS_SPATIAL, (64, 8, 8), 512, False),
    ("3x3 ds8 896->512", 4096, 896, ops.TAPS_SPATIAL, (64, 8, 8), 512, False),
    ("k3t ds8 512->512", 4096, 512, ops.TAPS_TEMPORAL, (16, 64, 1), 512, False),
    ("1x1 ds8 512->512 +res", 4096, 512, ops.TAPS_1, (1, 1, 1), 512, True),
    ("qkv ds8 512->1536", 4096, 512, ops.TAPS_1, (1, 1, 1), 1536, False),
    ("skip ds8 1024->512", 4096, 1024, ops.TAPS_1, (1, 1, 1), 512, False),
    ("3x3 ds4 384->384", 16384, 384, ops.TAPS_SPATIAL, (64, 16, 16), 384, False),
    ("3x3 ds4 896->384", 16384, 896, ops.TAPS_SPATIAL, (64, 16, 16), 384, False),
    ("k3t ds4 384->384", 16384, 384, ops.TAPS_TEMPORAL, (16, 256, 1), 384, False),
    ("1x1 ds4 384->384 +res", 16384, 384, ops.TAPS_1, (1, 1, 1), 384, True),
    ("skip ds4 768->384", 16384, 768, ops.TAPS_1, (1, 1, 1), 384, False),
    ("audio k3 ds8 512->512", 1600, 512, ops.taps_audio(2), (400, 1, 1), 512, False),
    ("audio k3 ds4 384->384", 6400, 384, ops.taps_audio(128), (1600, 1, 1), 384, False),
    ("audio k3 ds2 256->256", 25600, 256, ops.taps_audio(32), (6400, 1, 1), 256, False),
    ("3x3 ds2 256->256 (halo ref)", 65536, 256, ops.TAPS_SPATIAL, (64, 32, 32), 256, False),
]
TILES = (64, 129, 131, 132)


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
        ref = None
        line = f"{name:28s} M={M:6d} K={Cin*len(taps):5d} N={Cout:4d} tiles={-(-M//128)*-(-Cout//128):4d}"
        for tile in TILES:
            if tile == 131 and not ops.strip_tile_ok(x, Cout, taps):
                continue
            y = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile)
            if ref is None:
                ref = y.clone()
            same = torch.equal(y.view(torch.int16), ref.view(torch.int16))
            for _ in range(2):
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
            H.call("mmd_event_record", ev[0], st)
            n = 20
            for _ in range(n):
                ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
            H.call("mmd_event_record", ev[1], st)
            ms = ctypes.c_float()
            H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
            us = ms.value / n * 1000
            line += f" | t{tile}: {us:6.1f}us {flops/us/1e6:5.0f}TF {'=' if same else 'DIFF'}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
