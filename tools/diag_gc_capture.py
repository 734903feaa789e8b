#!/usr/bin/env python3
"""Root-cause probe for the round-1 SIGSEGV in `pytest -m gpu` (GraphStepper.step -> launch -> capture).

Hypothesis: model <-> engine reference cycles make UNetEngine.__del__ / GraphStepper.__del__ run from the cyclic GC at an arbitrary
allocation - e.g. while ANOTHER stepper is inside hipStreamBeginCapture..EndCapture - and the round-1 finalisers called
hipGraphExecDestroy there.  This script makes that interleaving deterministic: stepper A is captured, replayed and dropped (garbage,
not yet collected), then stepper B's capture runs with a gc.collect() forced in the middle of its launch plan.

Run from the root of the tree under test (round-1 tree: expected to die; current tree: must print OK)."""
import gc
import os
import sys

sys.path.insert(0, os.path.join(os.getcwd(), "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa: E402
from helpers import flags, synth_sd  # noqa: E402
from mm_diffusion import logger, multimodal_script_util as msu, ops  # noqa: E402
from mm_diffusion.sampler import GraphStepper  # noqa: E402

logger.set_quiet(True)
gc.disable()


def make():
    fl = flags("tiny", timestep_respacing="4")
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("tiny"))
    model.cuda().eval()
    st = GraphStepper(diff, model, 2, torch.device("cuda"))
    st.load(torch.randn(2, *fl["video_size"]).cuda(), torch.randn(2, *fl["audio_size"]).cuda())
    return model, diff, st


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for r in range(rounds):
        model, diff, st = make()
        st.step(3)
        st.step(2)
        torch.cuda.synchronize()
        del model, diff, st                      # a cycle: stays alive until the collector runs
        model, diff, st = make()
        real, calls = ops.run_plan, [0]

        def run_plan_with_gc(plan, stream, aux=None):
            calls[0] += 1
            if calls[0] == 3:                    # 1,2 = warm-up (U-Net plan, update plan); 3 = U-Net plan UNDER CAPTURE
                half = len(plan) // 2
                real(plan[:half], stream, aux)
                n = gc.collect()
                print(f"round {r}: gc.collect() in mid-capture freed {n} objects", flush=True)
                real(plan[half:], stream, aux)
            else:
                real(plan, stream, aux)

        ops.run_plan = run_plan_with_gc
        try:
            st.step(3)
        finally:
            ops.run_plan = real
        st.step(2)
        torch.cuda.synchronize()
        assert torch.isfinite(st.current()['video']).all()
        del model, diff, st
    print("OK: finalisers during capture were harmless")


if __name__ == "__main__":
    main()
