#!/usr/bin/env python3
"""Find a launch that reads memory it has no business reading (stale pool contents / uninitialised rows).

Two engines of the same model (identical plans, separate buffer pools) run the SAME inputs in lockstep, one launch at a time, from
different initial pool fills (zeros vs a small finite pattern).  Every byte a launch writes must be identical in both: the first launch
whose written region differs depends on what happened to be in memory before.  usage: stale_read_probe.py [tiny|mid|full] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, synth_sd  # noqa: E402
from mm_diffusion import _hip as H, logger, multimodal_script_util as msu  # noqa: E402

logger.set_quiet(True)
cfg = sys.argv[1] if len(sys.argv) > 1 else "full"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fl = flags(cfg, use_fp16=True, timestep_respacing="2")
model, diff = msu.create_model_and_diffusion(**fl)
model.load_state_dict(synth_sd("full" if cfg == "full" else "tiny"))
model.cuda().eval()
engs = [model.engine(B, torch.device("cuda"), replica=r) for r in range(2)]
g = torch.Generator().manual_seed(3)
video, audio = torch.randn(B, *fl["video_size"], generator=g).cuda(), torch.randn(B, *fl["audio_size"], generator=g).cuda()
t = torch.full((B,), 999, dtype=torch.int64).cuda()
shifts = [min(3, s) for s in model.draw_shifts()]


def pool_words(e):
    return torch.cat([r.view(torch.int16) for p in e.pools for r in p.all])


for k, e in enumerate(engs):
    for p in e.pools:
        for r in p.all:
            r.view(torch.int16).fill_(0 if k == 0 else 0x3c00)
    e.set_inputs(video, audio, t, shifts)
torch.cuda.synchronize()
assert len(engs[0].plan) == len(engs[1].plan)
lib = H.lib()
st = H.stream_handle()
bad = 0
for i, (oa, ob) in enumerate(zip(engs[0].plan, engs[1].plan)):
    if oa[0] is None:
        continue
    before = [pool_words(e).clone() for e in engs]
    for e, op in zip(engs, (oa, ob)):
        rc = op[0](*op[1], st)
        assert rc == 0, op[2]
    torch.cuda.synchronize()
    after = [pool_words(e) for e in engs]
    changed = (before[0] != after[0]) | (before[1] != after[1])
    diff_ = changed & (after[0] != after[1])
    nd = int(diff_.sum())
    if nd:
        bad += 1
        idx = diff_.nonzero()[:4].flatten().tolist()
        print(f"launch {i} {oa[2]} {oa[3][0] if oa[3] else ''} [{oa[5]}]: {nd} of {int(changed.sum())} written 16-bit words depend on stale memory, first word offsets {idx}",
              flush=True)
        if bad >= 6:
            break
print("launches with stale-memory dependence:", bad)
