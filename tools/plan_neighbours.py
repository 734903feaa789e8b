#!/usr/bin/env python3
"""Diagnostic (GPU): plan entries i0..i1 with their streams, and for a given pool buffer the buffers adjacent to it in the address space
with the plan entries that reference those.  usage: plan_neighbours.py <config> <buffer index> <i0> <i1>   (MMD_POOL_NOREUSE=1)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, inputs  # noqa: E402
from mm_diffusion import multimodal_script_util as msu  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402

name, target, i0, i1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
random.seed(5)
with torch.no_grad():
    model(v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda())
eng = next(iter(model._engines.values()))
raws = [r for p in eng.pools for r in p.all]
npool0 = len(eng.pools[0].all)
plan = [e for e in eng.plan if e[0] is not None]
spans = [(r.data_ptr(), r.data_ptr() + r.numel()) for r in raws]


def which(x):
    for bi, (lo, hi) in enumerate(spans):
        if lo <= x < hi:
            return bi, x - lo
    return None


for i in range(i0, i1):
    e = plan[i]
    ptrs = [(k, which(x)) for k, x in enumerate(e[1]) if isinstance(x, int) and x > (1 << 32)]
    print(i, "sid", e[4], e[2], e[3][0], [(k, w) for k, w in ptrs if w is not None])
order = sorted(range(len(raws)), key=lambda b: spans[b][0])
pos = order.index(target)
print("address neighbours of buffer", target)
for b in order[max(0, pos - 4): pos + 3]:
    lo, hi = spans[b]
    refs = [(i, e[2], e[3][0], e[4]) for i, e in enumerate(plan) if any(isinstance(x, int) and lo <= x < hi for x in e[1])]
    print(f"  buffer {b} pool {0 if b < npool0 else 1} [{lo:#x}, {hi:#x}) {hi - lo} B, gap to next {spans[order[order.index(b) + 1]][0] - hi if order.index(b) + 1 < len(order) else -1}: {refs[:4]}")
