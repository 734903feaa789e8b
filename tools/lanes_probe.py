#!/usr/bin/env python3
"""Probe: a 2-lane GraphStepper capture on the tiny model (MMD_LANE_FORK = raw | node | flat), with the native crash handler on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, synth_sd  # noqa: E402
from mm_diffusion import _hip as H, logger, multimodal_script_util as msu  # noqa: E402
from mm_diffusion.sampler import GraphStepper  # noqa: E402

H.lib().mmd_debug_install_crash_handler()
logger.set_quiet(True)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fl = flags("tiny", timestep_respacing="4")
model, diff = msu.create_model_and_diffusion(**fl)
model.load_state_dict(synth_sd("tiny"))
model.cuda().eval()
outs = []
for L in (1, lanes):
    st = GraphStepper(diff, model, 4, torch.device("cuda"), lanes=L)
    g = torch.Generator().manual_seed(5)
    st.load(torch.randn(4, *fl["video_size"], generator=g).cuda(), torch.randn(4, *fl["audio_size"], generator=g).cuda())
    import random
    random.seed(1)
    torch.manual_seed(1)
    for i in (3, 2, 1, 0):
        st.step(i)
    torch.cuda.synchronize()
    outs.append(st.current())
    print("lanes", L, "ok", flush=True)
print("equal:", torch.equal(outs[0]["video"], outs[1]["video"]), torch.equal(outs[0]["audio"], outs[1]["audio"]), "mode", os.environ.get("MMD_LANE_FORK", "node"))
