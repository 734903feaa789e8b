#!/usr/bin/env python3
"""Micro-benchmark of mmd_conv_wgrad (bf16) on training-step shapes at per-GPU batch 8.  MMD_WGRAD_TILE64=1 selects the
64x64 gather kernel instead of the 128x128 transposed-staging kernel."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

SHAPES = [("3x3 ds1 128->128", 524288, 128, ops.TAPS_SPATIAL, (128, 64, 64), 128), ("k3t ds1 128->128", 524288, 128, ops.TAPS_TEMPORAL, (16, 4096, 1), 128),
          ("3x3 ds2 256->256", 131072, 256, ops.TAPS_SPATIAL, (128, 32, 32), 256), ("qkv ds2 256->768", 131072, 256, ops.TAPS_1, (1, 1, 1), 768),
          ("3x3 ds4 384->384", 32768, 384, ops.TAPS_SPATIAL, (128, 16, 16), 384), ("3x3 ds8 512->512", 8192, 512, ops.TAPS_SPATIAL, (128, 8, 8), 512),
          ("1x1 ds1 128->128", 524288, 128, ops.TAPS_1, (1, 1, 1), 128), ("1x1 ds1 384->128", 524288, 384, ops.TAPS_1, (1, 1, 1), 128),
          ("1x1 ds2 256->256", 131072, 256, ops.TAPS_1, (1, 1, 1), 256), ("1x1 ds4 384->384", 32768, 384, ops.TAPS_1, (1, 1, 1), 384),
          ("1x1 ds8 512->512", 8192, 512, ops.TAPS_1, (1, 1, 1), 512), ("audio k3 128->128", 204800, 128, ops.taps_audio(4), (25600, 1, 1), 128),
          ("audio 1x1 256->256", 51200, 256, ops.TAPS_1, (1, 1, 1), 256), ("audio 1x1 512->512 L400", 3200, 512, ops.TAPS_1, (1, 1, 1), 512)]
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev:
    H.call("mmd_event_create", ctypes.byref(e))
st = H.stream_handle()
for name, M, Cin, taps, dims, Cout in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(M, Cout, device="cuda", generator=g).to(torch.bfloat16)
    dW = torch.zeros(Cout, Cin * len(taps), device="cuda")
    db = torch.zeros(Cout, device="cuda")
    for _ in range(2):
        ops.conv_wgrad(dy, x, dW, db, taps, dims)
    H.call("mmd_event_record", ev[0], st)
    n = 5
    for _ in range(n):
        ops.conv_wgrad(dy, x, dW, db, taps, dims)
    H.call("mmd_event_record", ev[1], st)
    ms = ctypes.c_float()
    H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
    us = ms.value / n * 1000
    fl = 2.0 * M * Cout * Cin * len(taps)
    print(f"{name:20s} {us:8.1f} us  {fl/us/1e6:6.0f} TF/s (wgrad + colsum)", flush=True)
