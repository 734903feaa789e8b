#!/usr/bin/env python3
"""What one dependent kernel node costs inside a replayed hipGraph (diagnostic, GPU): chains of N tiny launches (16-byte copy), captured
and replayed; also with a real mid-size kernel (gn_apply on 8 MB) between the tiny ones, to see the cost of a boundary next to real work."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

H.lib()
dev = torch.device("cuda")
a, b = torch.zeros(64, dtype=torch.uint8, device=dev), torch.zeros(64, dtype=torch.uint8, device=dev)
x = torch.randn(32768, 128, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
ga, gb = torch.ones(4, 128, device=dev), torch.zeros(4, 128, device=dev)
geom = ops.Geom.per_sample(4, 8192)
side = torch.cuda.Stream()


def chain(n, real):
    plan = []
    with ops.recording(plan):
        for i in range(n):
            if real:
                ops.gn_apply(x, ga, gb, geom, act=True, out=y)
            else:
                ops.copy2d(a.view(4, 16), b.view(4, 16))
    return plan


def timed(plan, reps=20):
    side.wait_stream(torch.cuda.current_stream())
    ops.run_plan(plan, side.cuda_stream)
    torch.cuda.synchronize()
    with H.capture(side.cuda_stream) as cap:
        ops.run_plan(plan, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    st = H.stream_handle()
    H.call("mmd_graph_launch", cap.exec, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        H.call("mmd_graph_launch", cap.exec, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (200, 800):
    t = timed(chain(n, False))
    print(f"{n} tiny dependent kernels in a graph: {t:.0f} us per replay = {t / n:.2f} us per node")
t1 = timed(chain(200, True))
print(f"200 gn_apply (8 MB in, 8 MB out) in a graph: {t1:.0f} us per replay = {t1 / 200:.2f} us per node; 16 MB at 4 TB/s = 4.2 us")
