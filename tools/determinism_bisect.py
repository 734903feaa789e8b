#!/usr/bin/env python3
"""Bisect the launch plan for the first entry whose presence makes graph replays differ (diagnostic, GPU).  For a prefix of k plan
entries (plus the join) a hipGraph is captured and replayed R times; after each replay every pool buffer is checksummed.  Binary
search for the smallest k whose replays are not all identical, then print the entries around it.
usage: determinism_bisect.py <config> <replays per probe>"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, inputs  # noqa: E402
from mm_diffusion import _hip as H, multimodal_script_util as msu, ops  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402

name, R = sys.argv[1], int(sys.argv[2])
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()
random.seed(5)
with torch.no_grad():
    model(v, a, t)
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))
raws = [r for p in eng.pools for r in p.all] + [eng.out_video.view(-1).view(torch.uint8), eng.out_audio.view(-1).view(torch.uint8)]
join = eng.join_plan()
plan = eng.plan
st = H.stream_handle()


SINGLE = os.environ.get("BISECT_SINGLE_STREAM") == "1"


def probe(k):
    """number of replays (of R) that differ from the first replay of the prefix plan[:k]"""
    sub = plan[:k] + join
    side = eng.side
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    with H.capture(side.cuda_stream) as cap:
        ops.run_plan(sub, side.cuda_stream, None if SINGLE else eng.aux.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)

    def once():
        random.seed(5)
        eng.set_inputs(v, a, t, model.draw_shifts())          # what model(v, a, t) does around the launch
        H.call("mmd_graph_launch", cap.exec, st)
        keep = (eng.out_video.clone(), eng.out_audio.clone())
        torch.cuda.synchronize()
        return torch.stack([r.view(torch.int32).sum() for r in raws]).cpu()
    once()
    ref = once()
    bad = sum(0 if torch.equal(once(), ref) else 1 for _ in range(R))
    H.retire("graph", cap.exec)
    return bad


full = probe(len(plan))
print(f"{name}: {len(plan)} entries; full plan: {full} of {R} replays differ")
if full:
    lo, hi = 0, len(plan)          # probe(lo) clean, probe(hi) dirty
    while hi - lo > 1:
        mid = (lo + hi) // 2
        b = probe(mid)
        print(f"  prefix {mid}: {b} of {R}")
        if b:
            hi = mid
        else:
            lo = mid
    print(f"first dirty prefix length {hi}: entry {hi - 1}")
    for j in range(max(0, hi - 12), min(len(plan), hi + 3)):
        e = plan[j]
        print("   ", j, e[2], e[3][0] if e[3] else e[1][:2], "sid", e[4])
    # confirm a few neighbours (the search assumes monotonicity)
    for k in (hi - 2, hi - 1, hi, hi + 1):
        if 0 < k <= len(plan):
            print(f"  recheck prefix {k}: {probe(k)} of {R}")
