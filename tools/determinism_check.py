#!/usr/bin/env python3
"""Run-to-run determinism of the launch plan (diagnostic, GPU): the same inputs through (a) one engine repeatedly, (b) a rebuilt engine
with the cached tile choices, (c) a rebuilt engine after clearing the autotuner's cache.  Prints which comparisons are bitwise and,
for (c), the launches whose tile choice changed."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, inputs  # noqa: E402
from mm_diffusion import multimodal_script_util as msu, ops  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mid"
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()


def run():
    random.seed(5)
    with torch.no_grad():
        ov, oa = model(v, a, t)
    return ov.clone(), oa.clone()


def same(x, y):
    return bool(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]))


def plan_tiles():
    eng = next(iter(model._engines.values()))
    return [(e[2], e[3][0] if e[3] else "") for e in eng.plan]


r = [run() for _ in range(4)]
print("one engine, runs 1-3 vs run 0:", [same(r[0], x) for x in r[1:]])
p0 = plan_tiles()
model.release_engines()
r2 = [run() for _ in range(2)]
print("rebuilt engine (cached tiles) vs first:", same(r[0], r2[0]), same(r[0], r2[1]))
model.release_engines()
ops._tile_cache.clear()
r3 = [run() for _ in range(2)]
p1 = plan_tiles()
print("rebuilt engine (re-tuned) vs first:", same(r[0], r3[0]), same(r[0], r3[1]), " self:", same(r3[0], r3[1]))
diff = [(i, x, y) for i, (x, y) in enumerate(zip(p0, p1)) if x != y]
print(f"{len(diff)} of {len(p0)} launches changed tile" + (" (plans differ in length)" if len(p0) != len(p1) else ""))
for d in diff[:12]:
    print("  ", d)
e = float((r3[0][0].float() - r[0][0].float()).norm() / r[0][0].float().norm())
print(f"rel-L2 video re-tuned vs first {e:.3e}")
