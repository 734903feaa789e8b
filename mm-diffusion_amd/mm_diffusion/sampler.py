"""Graph-replayed denoising step: [U-Net launch plan + fused DDPM update of both streams] captured once into a
hipGraph (mmd_graph_*), replayed per step.  Between replays only three small device buffers change: the
timestep (loop index + model timestep), the window shifts and the noise.

Replaces the per-step host work of the reference loop (gd:561-582 + resp:134-139): th.tensor([i]*B) H2D,
the timestep_map tensor rebuild, ~16 table uploads and ~55k ATen dispatches."""
import os

import torch as th

from . import _hip as H
from . import ops


def unwrap_unet(model):
    """Return the MultimodalUNet behind `model` (possibly wrapped by _WrappedModel / DDP), else None."""
    from .multimodal_unet import MultimodalUNet
    seen = 0
    while model is not None and seen < 4:
        if isinstance(model, MultimodalUNet):
            return model
        model = getattr(model, "model", None) or getattr(model, "module", None)
        seen += 1
    return None


def default_lanes(batch):
    """Batch lanes of a sampling step (see GraphStepper).  MMD_LANES overrides.  Round 6: TWO lanes for batches of 4 and more (each
    lane at least two samples).  Measured on MI355X at batch 4 (BASELINE configs[1]), same-call A/B on three boxes: 11.28 -> 11.09,
    10.85 -> 10.77, 10.87 -> 10.79 ms per step with two lanes (profiles/r06_suite_and_lanes_call9.txt, r06_wgrad_strip_lanes_call10.txt,
    r06_lanes_width_aconv_call11.txt); four lanes 15.0 ms (batch-1 launches fill a fraction of the chip and the 2 600 launches of a step
    queue).  Round 2 had measured 15.53 / 15.37 / 20.5 ms for 1 / 2 / 4 lanes and kept one: the per-level kernels have become short
    enough since that a second half-batch chain finds idle CUs beside the first one's latency-bound small levels.  The lanes are
    independent engines: the trajectory does not depend on the lane count (bitwise, tests/test_lifetime_gpu.py)."""
    v = os.environ.get("MMD_LANES")
    lanes = int(v) if v else (2 if batch >= 4 and batch % 2 == 0 else 1)
    return lanes if lanes >= 1 and batch % lanes == 0 else 1


class GraphStepper:
    """One denoising step of `batch` trajectories as one graph replay.

    lanes > 1 splits the batch into independent sub-batches, each with its own engine (activation buffers, video + audio launch
    streams; packed weights are shared) and its own captured graph; the lane graphs are launched on the lanes' private streams,
    forked from and joined to the caller's stream with events.  Every sample's trajectory is independent (GroupNorm / attention
    never mix batch elements, tests: batch sharding is bitwise exact), so the result is identical.  Noise and timestep buffers stay
    full-batch (lanes read slices), so the RNG stream does not depend on `lanes`.  Default: see default_lanes."""

    def __init__(self, diffusion, unet, batch, device, clip_denoised=True, use_graph=True, update="ddpm", eta=0.0, lanes=None):
        self.diff, self.unet, self.N = diffusion, unet, int(batch)
        self.device = th.device(device)
        self.lanes = default_lanes(self.N) if lanes is None else int(lanes)
        if self.N % self.lanes:
            raise H.MMDError(f"batch {self.N} does not split into {self.lanes} lanes")
        n = self.n = self.N // self.lanes
        self.engs = [unet.engine(n, self.device, replica=r) for r in range(self.lanes)]
        self.eng = e = self.engs[0]
        self.tab, _ = diffusion.device_tables(self.device)
        self.flags = diffusion._flags(clip_denoised)
        self.t_idx = th.zeros(self.N, dtype=th.int64, device=self.device)      # loop index (table row)
        self.noise_v = th.zeros((self.N,) + tuple(e.x_video.shape[1:]), dtype=th.float32, device=self.device)
        self.noise_a = th.zeros((self.N,) + tuple(e.x_audio.shape[1:]), dtype=th.float32, device=self.device)
        # model timestep: SpacedDiffusion maps the loop index to the original step (resp:134-139)
        tmap = getattr(diffusion, "timestep_map", None)
        self.tmap = list(tmap) if tmap is not None else list(range(diffusion.num_timesteps))
        self.rescale = bool(diffusion.rescale_timesteps)
        self.orig_T = getattr(diffusion, "original_num_steps", diffusion.num_timesteps)
        self.use_f32 = self.rescale
        F, C, HW = e.F, e.Cv_in, e.H0 * e.W0
        self.update_plans = []
        for r, e in enumerate(self.engs):
            sl = slice(r * n, (r + 1) * n)
            t_idx, nv, na = self.t_idx[sl], self.noise_v[sl], self.noise_a[sl]
            plan = []
            with ops.recording(plan):
                # in place: x_{t-1} overwrites x_t (purely elementwise); each update rides its own stream, then join
                ops.cur_sid = 0
                if update == "ddim":       # ddim_sample (gd:821-901): same graph, different fused update
                    tab3 = diffusion.ddim_tables(self.device)
                    ops.ddim_update(e.x_video, e.out_video, nv, e.x_video, self.tab, tab3, t_idx, F, C, HW, self.flags, eta)
                    ops.cur_sid = 1
                    ops.ddim_update(e.x_audio, e.out_audio, na, e.x_audio, self.tab, tab3, t_idx, 1, e.Ca_in, e.L0, self.flags, eta)
                else:
                    ops.ddpm_update(e.x_video, e.out_video, nv, e.x_video, self.tab, t_idx, F, C, HW, self.flags)
                    ops.cur_sid = 1
                    ops.ddpm_update(e.x_audio, e.out_audio, na, e.x_audio, self.tab, t_idx, 1, e.Ca_in, e.L0, self.flags)
                ops.cur_sid = 0
                ops.record_sync(1, 0)
            self.update_plans.append(plan)
        self.update_plan = self.update_plans[0]
        self.graphs = None
        self.use_graph = use_graph
        # per-step scalars go up through pinned rings (H.Staged): the host runs ahead of the GPU, one reused pinned buffer would race
        self._up_t = H.Staged(self.t_idx)
        self._up_tm = [H.Staged(e.t_f32 if self.use_f32 else e.t_i64) for e in self.engs]
        # lane fork / join events (lane 0 rides the origin stream, lane r > 0 its engine's private `side` stream)
        import ctypes
        self._lane_ev = []
        for _ in range(self.lanes + 1 if self.lanes > 1 else 0):
            ev = ctypes.c_void_p()
            H.call("mmd_event_create", ctypes.byref(ev))
            self._lane_ev.append(ev)

    def load(self, video, audio):
        n = self.n
        for r, e in enumerate(self.engs):
            e.x_video.copy_(video[r * n:(r + 1) * n])
            e.x_audio.copy_(audio[r * n:(r + 1) * n])

    def set_x(self, key, value):
        """Overwrite one stream of the current state (replacement-method conditional sampling)."""
        n = self.n
        for r, e in enumerate(self.engs):
            (e.x_video if key == "video" else e.x_audio).copy_(value[r * n:(r + 1) * n])

    def current(self):
        if self.lanes == 1:
            return {"video": self.eng.x_video.clone(), "audio": self.eng.x_audio.clone()}
        return {"video": th.cat([e.x_video for e in self.engs]), "audio": th.cat([e.x_audio for e in self.engs])}

    def _lane_launch(self, r, stream):
        """Enqueue lane r's U-Net plan and fused update with `stream` as its video-chain stream (its engine's aux = audio chain)."""
        e = self.engs[r]
        aux = e.aux.cuda_stream
        ops.run_plan(e.plan_f32 if self.use_f32 else e.plan, stream, aux)
        ops.run_plan(self.update_plans[r], stream, aux)

    def _capture(self):
        """One graph PER LANE, each captured with the lane's own private stream as the capture origin and its engine's aux stream
        as the only forked stream.  (All lanes in one capture would need two non-origin streams - a lane's video and audio chains
        - that wait on each other at every cross-attention; the HIP runtime re-parents the waiting stream on each such wait, the
        two become each other's parent and hipStreamEndCapture recurses until the stack overflows: the SIGSEGV of
        tools/lanes_probe.py.  Streams keep ONE role for life here: an engine's `side` is only ever an origin, its `aux` only ever
        the origin's child.)"""
        cur = th.cuda.current_stream(self.device)
        for r, e in enumerate(self.engs):
            e.side.wait_stream(cur)
            self._lane_launch(r, e.side.cuda_stream)        # warm-up: one-time function attributes, lazy module load
        th.cuda.synchronize(self.device)
        graphs = []
        for r, e in enumerate(self.engs):
            with H.capture(e.side.cuda_stream) as cap:
                self._lane_launch(r, e.side.cuda_stream)
            graphs.append(cap.exec)
        self.graphs = graphs

    def _fan(self, body):
        """Run body(r, stream) for every lane: lane streams fork from the current stream and join it again (plain events)."""
        cur = H.stream_handle()
        if self.lanes == 1:
            body(0, cur)
            return
        lib = H.lib()
        lib.mmd_event_record(self._lane_ev[0], cur)
        for r, e in enumerate(self.engs):
            ls = e.side.cuda_stream
            lib.mmd_stream_wait_event(ls, self._lane_ev[0])
            body(r, ls)
            lib.mmd_event_record(self._lane_ev[r + 1], ls)
        for r in range(self.lanes):
            if lib.mmd_stream_wait_event(cur, self._lane_ev[r + 1]):
                raise H.MMDError(f"lane join failed: {lib.mmd_last_error().decode()}")

    def set_step(self, i, shifts=None, noise=None):
        """Refresh the per-step device state: timestep, shifts, noise (video first, then audio - gd:453-454)."""
        self._up_t.host().fill_(int(i))
        self._up_t.push()
        tm = self.tmap[int(i)]
        for up in self._up_tm:
            up.host().fill_(float(tm) * (1000.0 / self.orig_T) if self.use_f32 else int(tm))
            up.push()
        shifts = self.unet.draw_shifts() if shifts is None else shifts      # one draw per block for the whole batch (unet:619-620)
        for e in self.engs:
            e.set_shifts(shifts)
        if noise is not None:
            self.noise_v.copy_(noise["video"])
            self.noise_a.copy_(noise["audio"])
        elif self.diff.noise_source is not None:
            self.noise_v.copy_(self.diff.noise_source(self.noise_v))
            self.noise_a.copy_(self.diff.noise_source(self.noise_a))
        else:
            self.noise_v.normal_()
            self.noise_a.normal_()

    def launch(self):
        if self.use_graph:
            if self.graphs is None:
                # capture replays the step once as warm-up: keep x intact around it
                keep = [(e.x_video.clone(), e.x_audio.clone()) for e in self.engs]
                self._capture()
                for e, (xv, xa) in zip(self.engs, keep):
                    e.x_video.copy_(xv)
                    e.x_audio.copy_(xa)
            self._fan(lambda r, stream: H.call("mmd_graph_launch", self.graphs[r], stream))
        else:
            if self.lanes == 1:
                self.eng.aux.wait_stream(th.cuda.current_stream(self.device))
            self._fan(self._lane_launch)

    def step(self, i, shifts=None, noise=None):
        self.set_step(i, shifts, noise)
        self.launch()

    def close(self):
        """Retire the graph exec and the update plan's join event (destroyed by H.reap() at the next safe point, never here: this
        also runs as a finaliser from the cyclic GC)."""
        try:
            gs, self.graphs = getattr(self, "graphs", None) or [], None
            for g in gs:
                H.retire("graph", g)
            for ev in H.plan_events(*(getattr(self, "update_plans", None) or [])) + list(getattr(self, "_lane_ev", [])):
                H.retire("event", ev)
            self.update_plans, self.update_plan, self._lane_ev = [], [], []
        except Exception:
            pass

    __del__ = close
