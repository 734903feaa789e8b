#!/usr/bin/env python3
"""Locate a run-to-run difference inside the launch plan (diagnostic, GPU).
  seq   : the plan replayed eagerly on ONE stream N times (no cross-stream concurrency) - how many runs differ from the first
  two   : the same on the engine's two streams (eager, not a graph)
  trace : one stream, and after EVERY launch an integer checksum of every pool buffer; reports the first launch after which a run
          differs from run 0 (kernel-internal races show up here with the launch's name)
usage: determinism_trace.py <config> <mode> <iters>"""
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

name, mode, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, 2, 3)
v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()
random.seed(5)
with torch.no_grad():
    model(v, a, t)
eng = next(iter(model._engines.values()))
random.seed(5)
shifts = [random.randint(0, 0) for _ in range(0)]
# the engine already holds inputs, timesteps and shifts of the call above: replay its plan directly
plan = eng.plan + eng.join_plan()
st = H.stream_handle()
raws = [r for p in eng.pools for r in p.all]
print(f"{len(plan)} plan entries, {len(raws)} pool buffers, {sum(r.numel() for r in raws) / 1e6:.1f} MB")


def outs():
    torch.cuda.synchronize()
    return eng.out_video.clone(), eng.out_audio.clone()


if mode in ("seq", "two"):
    def run():
        if mode == "seq":
            ops.run_plan(plan, st)
        else:
            eng.aux.wait_stream(torch.cuda.current_stream())
            ops.run_plan(plan, st, eng.aux.cuda_stream)
        return outs()
    ref = run()
    bad = [i for i in range(iters) if not all(torch.equal(x, y) for x, y in zip(run(), ref))]
    print(f"{name} {mode}: {len(bad)} of {iters} runs differ", bad[:10])
else:
    lib = H.lib()
    launches = [e for e in plan if e[0] is not None]

    def run():
        sums = []
        for fn, args, nm, _, sid, _tag in launches:
            rc = fn(*args, st)
            assert rc == 0, nm
            sums.append(torch.stack([r.view(torch.int32).sum() for r in raws]))
        return torch.stack(sums).cpu()
    ref = run()
    hits = {}
    for i in range(iters):
        s = run()
        d = (s != ref).any(dim=1).nonzero()
        if d.numel():
            k = int(d[0])
            buf = (s[k] != ref[k]).nonzero().flatten().tolist()
            hits.setdefault((k, launches[k][2], launches[k][3][0]), []).append((i, buf))
    print(f"{name} trace: {sum(len(v) for v in hits.values())} of {iters} runs differ")
    for k, v in sorted(hits.items()):
        print("  first differing launch", k, "runs", v[:5])
        i0 = k[0]
        for j in range(max(0, i0 - 3), min(len(launches), i0 + 2)):
            print("      ", j, launches[j][2], launches[j][3][0], "sid", launches[j][4])
