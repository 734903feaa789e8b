#!/usr/bin/env python3
"""Micro-benchmark of mmd_gn_bwd (bf16) on training-step shapes at per-GPU batch 8: time per call (zero + reduce + params + apply launches) and
the HBM rate of its compulsory traffic (reduce reads x, dy; apply reads x, dy, writes dx: 5 tensor passes)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

SHAPES = [("video ds1 128", 8, 65536, 128), ("video ds2 256", 8, 16384, 256), ("video ds4 384", 8, 4096, 384), ("video ds8 512", 8, 1024, 512),
          ("audio ds1 128", 8, 25600, 128), ("audio ds2 256", 8, 6400, 256), ("video ds1 256 (up)", 8, 65536, 256)]
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev:
    H.call("mmd_event_create", ctypes.byref(e))
st = H.stream_handle()
for name, S, Tn, C in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    M = S * Tn
    x = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    dx = torch.empty_like(x)
    gamma, beta = torch.randn(C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g)
    geom = ops.Geom.per_sample(S, Tn)
    mr = torch.empty(S, 32, 2, device="cuda")
    a, b = ops.gn_stats(x, gamma, beta, geom, mr=mr)
    dgamma, dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    for act in (True,):
        for _ in range(2):
            ops.gn_bwd(x, dy, dx, geom, a, b, mr, gamma, beta, None, act, dgamma, dbeta, None)
        H.call("mmd_event_record", ev[0], st)
        n = 10
        for _ in range(n):
            ops.gn_bwd(x, dy, dx, geom, a, b, mr, gamma, beta, None, act, dgamma, dbeta, None)
        H.call("mmd_event_record", ev[1], st)
        ms = ctypes.c_float()
        H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
        us = ms.value / n * 1000
        print(f"{name:22s} M={M:7d} C={C:4d} act={int(act)}  {us:8.1f} us   {5 * M * C * 2 / us / 1e6:6.2f} TB/s over 5 tensor passes", flush=True)
