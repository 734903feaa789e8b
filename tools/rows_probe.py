#!/usr/bin/env python3
"""Which rows of a batch-4 full-size 2-step run differ from the batch-1 runs of the same samples, by how much, and is either side
repeatable?  (diagnostic for tests/test_configs_gpu.py::test_config1_batch4_bf16_two_step_rows)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import flags, synth_sd  # noqa: E402
from mm_diffusion import logger, multimodal_script_util as msu  # noqa: E402
from mm_diffusion.sampler import GraphStepper  # noqa: E402

logger.set_quiet(True)
steps = int(os.environ.get("ROWS_STEPS", "2"))
fl = flags("full", use_fp16=True, timestep_respacing=str(steps))
model, diff = msu.create_model_and_diffusion(**fl)
model.load_state_dict(synth_sd("full"))
model.cuda().eval()
B = 4
g = torch.Generator().manual_seed(77)
xv, xa = torch.randn(B, *fl["video_size"], generator=g).cuda(), torch.randn(B, *fl["audio_size"], generator=g).cuda()
noise = [{"video": torch.randn(B, *fl["video_size"], generator=g).cuda(), "audio": torch.randn(B, *fl["audio_size"], generator=g).cuda()} for _ in range(steps)]
shifts = [[min(2 + k, 15) if i % 2 else 1 for i in range(15)] for k in range(steps)]


def run(rows, use_graph=True):
    st = GraphStepper(diff, model, len(rows), torch.device("cuda"), use_graph=use_graph)
    st.load(xv[rows], xa[rows])
    for k, i in enumerate(range(steps - 1, -1, -1)):
        st.step(i, shifts=shifts[k], noise={"video": noise[k]["video"][rows], "audio": noise[k]["audio"][rows]})
    out = st.current()
    st.close()
    return out


def d(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


full1, full2 = run([0, 1, 2, 3]), run([0, 1, 2, 3])
print("batch-4 repeatable:", torch.equal(full1["video"], full2["video"]), torch.equal(full1["audio"], full2["audio"]))
full3 = run([0, 1, 2, 3], use_graph=False)
print("batch-4 graph == eager:", torch.equal(full1["video"], full3["video"]), torch.equal(full1["audio"], full3["audio"]))
perm = run([3, 2, 1, 0])
print("batch-4 permuted rows equal:", [(torch.equal(perm["video"][3 - r], full1["video"][r]), torch.equal(perm["audio"][3 - r], full1["audio"][r])) for r in range(4)])
for order in ([0, 1, 2, 3], [3, 2, 1, 0]):
    for r in order:
        one = run([r])
        print(f"row {r}: video equal {torch.equal(one['video'][0], full1['video'][r])} rel {d(one['video'][0], full1['video'][r]):.2e} | "
              f"audio equal {torch.equal(one['audio'][0], full1['audio'][r])} rel {d(one['audio'][0], full1['audio'][r]):.2e}", flush=True)
two = run([1, 2])
print("batch-2 [1,2] vs batch-4 rows:", [(torch.equal(two["video"][i], full1["video"][1 + i]), torch.equal(two["audio"][i], full1["audio"][1 + i])) for i in range(2)])
