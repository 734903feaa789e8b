#!/usr/bin/env python3
"""Race detector for the two-stream plan: ONE engine (graph replay) is fed two different inputs A, B alternately, hundreds of times.  Every
out(A) must equal the first out(A) bitwise and every out(B) the first out(B).  The replay-determinism tests replay the SAME input, where a
launch that runs before its producer (a missing cross-stream wait) reads the previous replay's identical data and goes unnoticed; with
alternating inputs the previous replay's data is the OTHER input's.  usage: alternating_inputs_stress.py [tiny|mid|full] [batch] [iterations]
(the engine's A/B switches are read from the environment as usual)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, synth_sd  # noqa: E402
from mm_diffusion import logger, multimodal_script_util as msu  # noqa: E402


def run(cfg="mid", B=1, iters=300, verbose=True):
    logger.set_quiet(True)
    fl = flags(cfg, use_fp16=True)
    model, _ = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("full" if cfg == "full" else "tiny"))
    model.cuda().eval()
    g = torch.Generator().manual_seed(11)
    ins = []
    for k in range(2):
        ins.append((torch.randn(B, *fl["video_size"], generator=g).cuda(), torch.randn(B, *fl["audio_size"], generator=g).cuda(),
                    torch.full((B,), 17 + 383 * k, dtype=torch.int64).cuda(), 1 + 2 * k))
    ref = [None, None]
    bad = [0, 0]
    worst = 0.0
    for it in range(iters):
        k = it & 1
        v, a, t, sh = ins[k]
        model.shift_source = lambda lo, hi, sh=sh: min(sh, hi)
        with torch.no_grad():
            ov, oa = model(v, a, t)
        if ref[k] is None:
            ref[k] = (ov.clone(), oa.clone())
        elif not (torch.equal(ov, ref[k][0]) and torch.equal(oa, ref[k][1])):
            bad[k] += 1
            worst = max(worst, float((ov - ref[k][0]).norm() / ref[k][0].norm()), float((oa - ref[k][1]).norm() / ref[k][1].norm()))
    if verbose:
        print(f"{cfg} batch {B}: {iters} alternating replays: {bad[0]} + {bad[1]} differ"
              + (f" (worst rel-L2 {worst:.2e})" if sum(bad) else ""))
    return sum(bad), worst


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "mid"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    sys.exit(1 if run(cfg, B, iters)[0] else 0)
