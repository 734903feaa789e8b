#!/usr/bin/env python3
"""Run one mmd_conv_gemm shape/tile N times (for rocprofv3 --pmc passes).  usage: gemm_one.py <shape-index> <tile> [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from mm_diffusion import ops  # noqa: E402
from gemm_bench import SHAPES  # noqa: E402

name, M, Cin, taps, dims, Cout, res = SHAPES[int(sys.argv[1])]
tile = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(dt)
b = torch.randn(Cout, device="cuda", generator=g)
r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
y = torch.empty(M, Cout, device="cuda", dtype=dt)
for _ in range(reps):
    ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
torch.cuda.synchronize()
print(name, tile, "done")
