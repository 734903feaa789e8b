#!/usr/bin/env python3
"""Diagnostic (GPU): replay a hand-picked pair of kernel chains from the recorded plan - a video-stream chain and an audio-stream chain -
concurrently inside one hipGraph, many times, and watch the video chain's outputs for differences.  Run with MMD_POOL_NOREUSE=1 so that
every intermediate of the full forward is still intact in its own buffer (the chains read real data).
usage: determinism_mini.py <config> <video entries, comma separated> <audio entries> <reps inside the graph> <replays>"""
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

name = sys.argv[1]
vids = [int(x) for x in sys.argv[2].split(",") if x]
auds = [int(x) for x in sys.argv[3].split(",") if x]
reps, R = int(sys.argv[4]), int(sys.argv[5])
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
random.seed(5)
with torch.no_grad():
    model(v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda())
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))
plan = [e for e in eng.plan if e[0] is not None]
raws = [r for p in eng.pools for r in p.all]
spans = [(r.data_ptr(), r.data_ptr() + r.numel()) for r in raws]
touched = sorted({bi for i in vids + auds for x in plan[i][1] if isinstance(x, int) for bi, (lo, hi) in enumerate(spans) if lo <= x < hi})
fork, join = [], []
with ops.recording(fork):
    ops.record_sync(0, 1)
with ops.recording(join):
    ops.record_sync(1, 0)
mini = list(fork)
for _ in range(reps):
    mini += [plan[i][:4] + (0, "") for i in vids]
    mini += [plan[i][:4] + (1, "") for i in auds]
mini += join
side = eng.side
side.wait_stream(torch.cuda.current_stream())
ops.run_plan(mini, side.cuda_stream, eng.aux.cuda_stream)
torch.cuda.synchronize()
with H.capture(side.cuda_stream) as cap:
    ops.run_plan(mini, side.cuda_stream, eng.aux.cuda_stream)
torch.cuda.current_stream().wait_stream(side)
st = H.stream_handle()


def once():
    H.call("mmd_graph_launch", cap.exec, st)
    torch.cuda.synchronize()
    return torch.stack([raws[b].view(torch.int32).sum() for b in touched]).cpu()


good = {b: raws[b].clone() for b in touched}          # the full forward's (and the eager warm-up's) contents
ref = once()
for b in touched:
    if not torch.equal(raws[b], good[b]):
        print(f"  first replay already differs from the eager contents in buffer {b}")
bad = {}
shown = 0
for i in range(R):
    s = once()
    d = tuple(touched[k] for k in (s != ref).nonzero().flatten().tolist())
    if d:
        bad[d] = bad.get(d, 0) + 1
        if shown < 3:
            shown += 1
            for b in d[:1]:
                x, g = raws[b].view(torch.bfloat16).float(), good[b].view(torch.bfloat16).float()
                idx = (x != g).nonzero().flatten()
                C = int(os.environ.get("MINI_C", "128"))
                rows, cols = (idx // C), (idx % C)
                print(f"  replay {i}: buffer {b}: {idx.numel()} of {x.numel()} elements differ; rows {sorted(set(rows.tolist()))[:24]} ... cols {sorted(set(cols.tolist()))[:40]};"
                      f" max |diff| {float((x - g).abs().max()):.3e}; sample got/want {[(float(x[j]), float(g[j])) for j in idx[:4].tolist()]}")
print(f"video {vids} x audio {auds}, {reps} reps per graph: {sum(bad.values())} of {R} replays differ; buffers {bad}")
