#!/usr/bin/env python3
"""Do the two batch lanes of a sampling step gain from running OUT OF PHASE?  (timing only)

The product joins the lanes at every step (sampler.GraphStepper._fan), so both lanes run the same U-Net level at the same time: two
chip-filling ds1 launches time-share, two latency-bound ds8 launches leave the chip idle together.  This probe replays the captured
per-lane graphs of BASELINE configs[1] (batch 4 = 2 lanes x 2 samples) with frozen per-step inputs (timestep, shifts, noise: the kernels'
time does not depend on them)
    joined   : fork / join per step, the product's schedule
    free     : no joins - each lane's stream replays its K steps back to back
    skew f   : lane 1 starts f of a step after lane 0, then free
and prints wall time per step (all lanes finish K steps; the skew's start-up is included).  usage: lane_skew_probe.py [K] [lanes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from mm_diffusion import _hip as H  # noqa: E402
from mm_diffusion.sampler import GraphStepper  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    fl, model, diff = bench.build("bf16", "250", 4, dev)
    torch.manual_seed(0)
    st = GraphStepper(diff, model, 4, dev, clip_denoised=True, lanes=lanes)
    st.load(torch.randn(4, *fl["video_size"]).to(dev), torch.randn(4, *fl["audio_size"]).to(dev))
    T = diff.num_timesteps
    for i in range(3):
        st.step(T - 1 - i)
    torch.cuda.synchronize()
    streams = [e.side.cuda_stream for e in st.engs]

    def joined():
        for _ in range(K):
            st.launch()

    def free(skew_ms=0.0):
        H.call("mmd_graph_launch", st.graphs[0], streams[0])
        if skew_ms > 0:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e3 < skew_ms:
                pass
        for k in range(K):
            for r in range(lanes):
                if r == 0 and k == K - 1:
                    continue                     # lane 0's first step went out ahead
                H.call("mmd_graph_launch", st.graphs[r], streams[r])

    def timed(fn, *a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(*a)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / K

    def alone(r):
        for _ in range(K):
            H.call("mmd_graph_launch", st.graphs[r], streams[r])

    for r in range(lanes):
        print(f"lane {r} ALONE (its graph replayed K times, the other lane idle): {min(timed(alone, r) for _ in range(2)):.3f} ms per lane-step", flush=True)
    base = min(timed(joined) for _ in range(2))
    print(f"K = {K} steps, {lanes} lanes; joined (product schedule, inputs frozen): {base:.3f} ms per step", flush=True)
    for rep in range(3):
        line = [f"pass {rep}: joined {timed(joined):.3f}", f"free {timed(free):.3f}"]
        for f in (0.15, 0.3, 0.5, 0.7):
            line.append(f"skew {f:.2f} {timed(free, f * base):.3f}")
        print(" | ".join(line), flush=True)


if __name__ == "__main__":
    main()
