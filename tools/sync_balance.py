#!/usr/bin/env python3
"""Who waits for whom at the cross-modal sync points (diagnostic, GPU).  Every launch of the configs[1] plan is timed in isolation (HIP
events, one stream); the two-stream schedule is then replayed on paper: a launch advances its stream's clock by its isolated time, a
sync marker (src -> dst) lifts dst's clock to src's.  Printed: each marker where a stream would wait, the total wait per stream, the
paper makespan against the sum of the video stream's kernels - i.e. how much of the step the audio chain costs even with a perfect
machine, before any slowdown from sharing the CUs."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from mm_diffusion import _hip as H  # noqa: E402
from mm_diffusion.sampler import GraphStepper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
fl, model, diff = bench.build("bf16", "250", B, dev)
st = GraphStepper(diff, model, B, dev, clip_denoised=True, lanes=1)
st.load(torch.randn(B, *fl["video_size"]).to(dev), torch.randn(B, *fl["audio_size"]).to(dev))
for i in (249, 248):
    st.step(i)
torch.cuda.synchronize()
full = (st.eng.plan_f32 if st.use_f32 else st.eng.plan) + st.update_plan
launches = [e for e in full if e[0] is not None]
lib, stream = H.lib(), H.stream_handle()
evs = []
for _ in range(len(launches) + 1):
    e = ctypes.c_void_p()
    H.call("mmd_event_create", ctypes.byref(e))
    evs.append(e)
best = [1e9] * len(launches)
for rep in range(5):
    torch.cuda.synchronize()
    lib.mmd_event_record(evs[0], stream)
    for i, e in enumerate(launches):
        assert e[0](*e[1], stream) == 0
        lib.mmd_event_record(evs[i + 1], stream)
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    for i in range(len(launches)):
        H.call("mmd_event_elapsed_ms", evs[i], evs[i + 1], ctypes.byref(ms))
        best[i] = min(best[i], ms.value * 1e3)
FLOOR = float(os.environ.get("SB_FLOOR_US", "4.0"))       # the event-to-event floor of an empty launch from Python; a graph node costs ~1.5 us
dur = [max(b - FLOOR, 0.0) + 1.5 for b in best]
clock, busy, wait = [0.0, 0.0], [0.0, 0.0], [0.0, 0.0]
k = 0
print(f"{len(launches)} launches, {sum(1 for e in full if e[0] is None)} sync markers; isolated kernel sums: video {sum(d for d, e in zip(dur, launches) if e[4] == 0) / 1e3:.2f} ms, "
      f"audio {sum(d for d, e in zip(dur, launches) if e[4] == 1) / 1e3:.2f} ms")
for e in full:
    if e[0] is None:
        src, dst, _ = e[1]
        if clock[src] > clock[dst] + 1.0:
            w = clock[src] - clock[dst]
            wait[dst] += w
            nxt = next((x[3][0] for x in full[full.index(e) + 1:] if x[0] is not None and x[4] == dst), "")
            print(f"  at launch {k:4d}: stream {dst} waits {w:7.1f} us for stream {src} (clocks {clock[dst] / 1e3:.2f} / {clock[src] / 1e3:.2f} ms), next on {dst}: {nxt[:70]}")
        clock[dst] = max(clock[dst], clock[src])
    else:
        clock[e[4]] += dur[k]
        busy[e[4]] += dur[k]
        k += 1
print(f"paper makespan {max(clock) / 1e3:.2f} ms; video busy {busy[0] / 1e3:.2f} ms + waits {wait[0] / 1e3:.2f} ms; audio busy {busy[1] / 1e3:.2f} ms + waits {wait[1] / 1e3:.2f} ms")
