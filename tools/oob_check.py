#!/usr/bin/env python3
"""Out-of-bounds store detector for the recorded plan (diagnostic, GPU; run with MMD_POOL_NOREUSE=1).  After one forward every tensor
sits intact in its own buffer.  For each distinct launch of the plan: every pool buffer the launch does NOT reference is filled with
a pattern, the launch runs alone, and the pattern is checked - a changed byte outside the referenced buffers is a stray store.
usage: oob_check.py <config> [batch]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, inputs  # noqa: E402
from mm_diffusion import _hip as H, multimodal_script_util as msu  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fl = flags(name, use_fp16=True)
model, _ = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().eval()
v, a = inputs(fl, B, 3)
random.seed(5)
with torch.no_grad():
    model(v.cuda(), a.cuda(), torch.tensor([17, 400, 3, 999][:B]).cuda())
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))
plan = [e for e in eng.plan if e[0] is not None]
raws = [r for p in eng.pools for r in p.all]
spans = [(r.data_ptr(), r.data_ptr() + r.numel()) for r in raws]
snap = [r.clone() for r in raws]
PAT = 0x5A
st = H.stream_handle()
seen, nbad = set(), 0
for i, e in enumerate(plan):
    key = (e[2], e[3][0])
    if key in seen:
        continue
    seen.add(key)
    refs = {bi for x in e[1] if isinstance(x, int) for bi, (lo, hi) in enumerate(spans) if lo <= x < hi}
    others = [b for b in range(len(raws)) if b not in refs]
    for b in others:
        raws[b].fill_(PAT)
    rc = e[0](*e[1], st)
    assert rc == 0, key
    torch.cuda.synchronize()
    hit = [b for b in others if bool((raws[b] != PAT).any())]
    if hit:
        nbad += 1
        for b in hit[:4]:
            idx = (raws[b] != PAT).nonzero().flatten()
            print(f"STRAY STORE: entry {i} {key} sid {e[4]} wrote {idx.numel()} bytes of buffer {b} ({raws[b].numel()} B) at offsets {idx[:4].tolist()} .. {idx[-1].item()};"
                  f" referenced buffers {sorted(refs)}; buffer {b} lies {spans[b][0] - max(spans[r][1] for r in refs):+d} B after the end of the last referenced one")
    for b in others:
        raws[b].copy_(snap[b])
print(f"{name} batch {B}: {len(seen)} distinct launches checked, {nbad} with stray stores")
