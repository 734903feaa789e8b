#!/usr/bin/env python3
"""Where does the host time of the graph-captured training step go?  cProfile of two steps after capture (bench.py --mode train shapes)."""
import cProfile
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mm_diffusion import logger, multimodal_script_util as msu  # noqa: E402
from mm_diffusion.optim import FlatAdamW  # noqa: E402
from mm_diffusion.synth import synth_init_  # noqa: E402
from mm_diffusion.train_graph import GraphedTrainStep  # noqa: E402

logger.set_quiet(True)
fl = msu.model_and_diffusion_defaults()
fl.update(bench.FULL)
fl.update(use_fp16=True, dropout=0.1)
model, diff = msu.create_model_and_diffusion(**fl)
synth_init_(model)
model.cuda().train()
opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=0.0, ema_rates=[0.9999], pack_dtype=model.dtype)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(1)
random.seed(1)
x0 = {"video": (torch.rand(B, *fl["video_size"], generator=g) * 2 - 1).cuda(), "audio": (torch.rand(B, *fl["audio_size"], generator=g) * 2 - 1).cuda()}
gs = GraphedTrainStep(model, diff, opt, x0)


def step():
    t = torch.randint(0, diff.num_timesteps, (B,), generator=g).cuda()
    return gs.step(x0, t)["loss"].mean()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"3 steps: host {1000 * h / 3:.1f} ms/step, wall {1000 * (time.perf_counter() - t0) / 3:.1f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
