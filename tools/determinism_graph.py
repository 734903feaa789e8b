#!/usr/bin/env python3
"""Which buffers differ after a graph replay that differs (diagnostic, GPU).  After every replay of the captured plan the checksum of
every pool buffer is compared with the first replay's; for a differing replay, the differing buffers are listed with the plan entries
that reference them (index, entry point, label, stream) - the earliest of those is where the corruption entered.
usage: determinism_graph.py <config> <iters>"""
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

name, iters = sys.argv[1], int(sys.argv[2])
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()


def run():
    random.seed(5)
    with torch.no_grad():
        model(v, a, t)
    torch.cuda.synchronize()
    return torch.stack([r.view(torch.int32).sum() for r in raws]).cpu()


random.seed(5)
with torch.no_grad():
    model(v, a, t)
eng = next(iter(model._engines.values()))
raws = [r for p in eng.pools for r in p.all]
spans = [(r.data_ptr(), r.data_ptr() + r.numel()) for r in raws]
plan = [e for e in eng.plan if e[0] is not None]


def refs(bi):
    lo, hi = spans[bi]
    return [(i, e[2], e[3][0], e[4], [k for k, x in enumerate(e[1]) if isinstance(x, int) and lo <= x < hi]) for i, e in enumerate(plan)
            if any(isinstance(x, int) and lo <= x < hi for x in e[1])]


ref = run()
seen = {}
nbad = 0
for i in range(iters):
    s = run()
    d = (s != ref).nonzero().flatten().tolist()
    if d:
        nbad += 1
        seen.setdefault(tuple(d), []).append(i)
print(f"{name}: {nbad} of {iters} replays differ; {len(raws)} buffers; patterns: { {k: len(v) for k, v in seen.items()} }")
npool0 = len(eng.pools[0].all)
for pat in list(seen)[:2]:
    bad = set(pat)
    print("pattern of", len(pat), "buffers;", seen[pat][:5])
    # buffers are allocated in plan order per pool: show the neighbourhood of the first differing buffer that is followed by a run of
    # differing ones (skip-concat buffers allocated early are written late and differ anyway)
    for lo, hi in ((0, npool0), (npool0, len(raws))):
        run = next((b for b in range(lo, hi - 3) if all(x in bad for x in range(b, b + 4))), None)
        if run is None:
            continue
        for bi in range(max(lo, run - 3), min(hi, run + 4)):
            rr = refs(bi)
            print(f"  buffer {bi} {'DIFF' if bi in bad else 'same'} ({raws[bi].numel()} B, pool {0 if bi < npool0 else 1}) referenced by {len(rr)} entries:")
            for r in rr[:6]:
                print("     ", r)
