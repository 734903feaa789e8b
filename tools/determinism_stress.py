#!/usr/bin/env python3
"""Stress the replayed plan for run-to-run differences (diagnostic, GPU): N forwards of one engine on the same inputs, optionally while
a second process keeps the GPU busy (timing perturbation, like two ranks sharing a device).  Prints the number of replays whose output
differs from the first one.  usage: determinism_stress.py <config> <iters> [noise]"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "--noise":
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(64 << 20, device="cuda")
    import time
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            a @ a
            x.mul_(1.0001)
        torch.cuda.synchronize()
    sys.exit(0)

from helpers import flags, inputs  # noqa: E402
from mm_diffusion import multimodal_script_util as msu  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402

name, iters = sys.argv[1], int(sys.argv[2])
noise = len(sys.argv) > 3
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
B = 2
v, a = inputs(fl, B, 3)
v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()


def run():
    random.seed(5)
    with torch.no_grad():
        ov, oa = model(v, a, t)
    return ov, oa


ref = run()
proc = subprocess.Popen([sys.executable, __file__, "--noise", "60"]) if noise else None
bad = []
for i in range(iters):
    o = run()
    if not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])):
        ev = float((o[0].float() - ref[0].float()).norm() / ref[0].float().norm())
        ea = float((o[1].float() - ref[1].float()).norm() / ref[1].float().norm())
        bad.append((i, ev, ea))
if proc:
    proc.kill()
print(f"{name} noise={noise} env={ {k: v for k, v in os.environ.items() if k.startswith('MMD_')} }: {len(bad)} of {iters} replays differ", bad[:6])
