"""Process-lifetime behaviour of the HIP path: finalisers, captures, the host running ahead of the GPU.

Round-1's `pytest -m gpu` died with a host SIGSEGV inside GraphStepper.step -> launch (capture): engines and steppers sit in
reference cycles, so their finalisers ran from the cyclic GC at arbitrary points and destroyed graph execs while another capture
was open.  The rules tested here: finalisers only RETIRE handles (mm_diffusion._hip.retire / reap), captures run with the GC off on
a private stream, per-step scalars go up through pinned rings."""
import gc
import random

import pytest
import torch

from helpers import flags, synth_sd

pytestmark = pytest.mark.gpu


def _make(resp="4", B=2, dt=torch.float32):
    from mm_diffusion import logger, multimodal_script_util as msu
    from mm_diffusion.sampler import GraphStepper
    logger.set_quiet(True)
    fl = flags("tiny", timestep_respacing=resp, use_fp16=(dt == torch.bfloat16))
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("tiny"))
    model.cuda().eval()
    st = GraphStepper(diff, model, B, torch.device("cuda"))
    g = torch.Generator().manual_seed(5)
    st.load(torch.randn(B, *fl["video_size"], generator=g).cuda(), torch.randn(B, *fl["audio_size"], generator=g).cuda())
    return fl, model, diff, st


def test_finalisers_during_capture_are_harmless():
    """Garbage engine + stepper (captured graphs, events, streams) collected in the MIDDLE of another stepper's capture."""
    from mm_diffusion import _hip as H, ops
    was = gc.isenabled()
    gc.disable()
    try:
        fl, model, diff, st = _make()
        st.step(3)
        st.step(2)
        torch.cuda.synchronize()
        del model, diff, st                                   # cycle: alive until the collector runs
        fl, model, diff, st = _make()
        real, calls, freed = ops.run_plan, [0], [0]

        def run_plan_with_gc(plan, stream, aux=None):
            calls[0] += 1
            if calls[0] == 3:                                 # 1, 2 = warm-up plans; 3 = the U-Net plan under capture
                half = len(plan) // 2
                real(plan[:half], stream, aux)
                assert H._capturing == 1
                freed[0] = gc.collect()
                real(plan[half:], stream, aux)
            else:
                real(plan, stream, aux)

        ops.run_plan = run_plan_with_gc
        try:
            st.step(3)
        finally:
            ops.run_plan = real
        assert freed[0] > 0 and len(H._retired) > 0           # the finalisers ran mid-capture and only retired their handles
        st.step(2)
        torch.cuda.synchronize()
        cur = st.current()
        assert torch.isfinite(cur["video"]).all() and torch.isfinite(cur["audio"]).all()
        assert H.reap() > 0 and not H._retired                # destroyed at a safe point
    finally:
        if was:
            gc.enable()


def test_capture_result_unaffected_by_finalisers():
    """Same trajectory with and without a collection in mid-capture: bitwise equal."""
    from mm_diffusion import ops
    outs = []
    for inject in (False, True):
        if inject:
            junk = _make()                                    # noqa: F841  garbage with live HIP handles
            junk[3].step(3)
            del junk
        fl, model, diff, st = _make()
        random.seed(11)
        torch.manual_seed(11)
        real, calls = ops.run_plan, [0]

        def hooked(plan, stream, aux=None):
            calls[0] += 1
            real(plan, stream, aux)
            if inject and calls[0] == 3:
                gc.collect()

        ops.run_plan = hooked
        try:
            for i in (3, 2, 1, 0):
                st.step(i)
        finally:
            ops.run_plan = real
        cur = st.current()
        outs.append((cur["video"], cur["audio"]))
        st.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_long_graph_replay_equals_synchronous_eager_loop(dt):
    """250 steps with DEVICE noise: the graph-replay loop (host free-running many steps ahead of the GPU) must equal the eager loop
    with a device sync after every step, bitwise - a reused pinned staging buffer for (t, shifts) would show up here as a
    timestep / window shift of a later step leaking into an earlier one."""
    finals = []
    for use_graph in (True, False):
        fl, model, diff, st = _make(resp="250", B=2, dt=dt)
        st.use_graph = use_graph
        random.seed(3)
        torch.manual_seed(3)
        for i in range(diff.num_timesteps - 1, -1, -1):
            st.step(i)
            if not use_graph:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        cur = st.current()
        finals.append((cur["video"], cur["audio"]))
        st.close()
    assert torch.isfinite(finals[0][0]).all() and torch.isfinite(finals[0][1]).all()
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])


def test_staged_upload_ring_orders_with_the_stream():
    """H.Staged: 64 back-to-back uploads of different values, each consumed by a stream-ordered device copy, without any host sync."""
    from mm_diffusion import _hip as H
    dev = torch.zeros(8, dtype=torch.int64, device="cuda")
    up = H.Staged(dev, depth=4)
    seen = torch.zeros(64, 8, dtype=torch.int64, device="cuda")
    big = torch.randn(1 << 24, device="cuda")
    for i in range(64):
        up.host().fill_(i)
        up.push()
        seen[i].copy_(dev)
        big.mul_(1.0001)                                      # keep the GPU busy so the host runs ahead
    torch.cuda.synchronize()
    assert torch.equal(seen.cpu(), torch.arange(64).view(64, 1).expand(64, 8))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batch_lanes_do_not_change_the_trajectory(dt):
    """GraphStepper lanes: the batch split into 1 / 2 / 4 concurrently replayed sub-batches gives bitwise the same samples
    (independent trajectories; noise and timestep buffers are full-batch, so the RNG stream does not depend on the split)."""
    outs = []
    for lanes in (1, 2, 4):
        from mm_diffusion.sampler import GraphStepper
        fl, model, diff, st0 = _make(resp="4", B=4, dt=dt)
        st0.close()
        st = GraphStepper(diff, model, 4, torch.device("cuda"), lanes=lanes)
        g = torch.Generator().manual_seed(5)
        st.load(torch.randn(4, *fl["video_size"], generator=g).cuda(), torch.randn(4, *fl["audio_size"], generator=g).cuda())
        random.seed(21)
        torch.manual_seed(21)
        for i in (3, 2, 1, 0):
            st.step(i)
        cur = st.current()
        outs.append((cur["video"], cur["audio"]))
        assert len(st.engs) == lanes
        st.close()
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
