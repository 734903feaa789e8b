"""Graph-replayed denoising step: [U-Net launch plan + fused DDPM update of both streams] captured once into a
hipGraph (mmd_graph_*), replayed per step.  Between replays only three small device buffers change: the
timestep (loop index + model timestep), the window shifts and the noise.

Replaces the per-step host work of the reference loop (gd:561-582 + resp:134-139): th.tensor([i]*B) H2D,
the timestep_map tensor rebuild, ~16 table uploads and ~55k ATen dispatches."""
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


class GraphStepper:
    def __init__(self, diffusion, unet, batch, device, clip_denoised=True, use_graph=True, update="ddpm", eta=0.0):
        self.diff, self.unet, self.N = diffusion, unet, int(batch)
        self.device = th.device(device)
        self.eng = unet.engine(self.N, self.device)
        e = self.eng
        self.tab, _ = diffusion.device_tables(self.device)
        self.flags = diffusion._flags(clip_denoised)
        self.t_idx = th.zeros(self.N, dtype=th.int64, device=self.device)      # loop index (table row)
        self.noise_v = th.zeros_like(e.x_video)
        self.noise_a = th.zeros_like(e.x_audio)
        # model timestep: SpacedDiffusion maps the loop index to the original step (resp:134-139)
        tmap = getattr(diffusion, "timestep_map", None)
        self.tmap = list(tmap) if tmap is not None else list(range(diffusion.num_timesteps))
        self.rescale = bool(diffusion.rescale_timesteps)
        self.orig_T = getattr(diffusion, "original_num_steps", diffusion.num_timesteps)
        self.use_f32 = self.rescale
        F, C, HW = e.F, e.Cv_in, e.H0 * e.W0
        self.update_plan = []
        with ops.recording(self.update_plan):
            # in place: x_{t-1} overwrites x_t (purely elementwise); each update rides its own stream, then join
            ops.cur_sid = 0
            if update == "ddim":       # ddim_sample (gd:821-901): same graph, different fused update
                tab3 = diffusion.ddim_tables(self.device)
                ops.ddim_update(e.x_video, e.out_video, self.noise_v, e.x_video, self.tab, tab3, self.t_idx, F, C, HW, self.flags, eta)
                ops.cur_sid = 1
                ops.ddim_update(e.x_audio, e.out_audio, self.noise_a, e.x_audio, self.tab, tab3, self.t_idx, 1, e.Ca_in, e.L0, self.flags, eta)
            else:
                ops.ddpm_update(e.x_video, e.out_video, self.noise_v, e.x_video, self.tab, self.t_idx, F, C, HW, self.flags)
                ops.cur_sid = 1
                ops.ddpm_update(e.x_audio, e.out_audio, self.noise_a, e.x_audio, self.tab, self.t_idx, 1, e.Ca_in, e.L0, self.flags)
            ops.cur_sid = 0
            ops.record_sync(1, 0)
        self.graph = None
        self.use_graph = use_graph
        self._host_t = th.zeros(self.N, dtype=th.int64).pin_memory()
        self._host_tm = (th.zeros(self.N, dtype=th.float32) if self.use_f32 else th.zeros(self.N, dtype=th.int64)).pin_memory()

    def load(self, video, audio):
        self.eng.x_video.copy_(video)
        self.eng.x_audio.copy_(audio)

    def current(self):
        return {"video": self.eng.x_video.clone(), "audio": self.eng.x_audio.clone()}

    def _launch_all(self, stream):
        aux = self.eng.aux.cuda_stream
        ops.run_plan(self.eng.plan_f32 if self.use_f32 else self.eng.plan, stream, aux)
        ops.run_plan(self.update_plan, stream, aux)

    def _capture(self):
        side = th.cuda.Stream(device=self.device)
        side.wait_stream(th.cuda.current_stream(self.device))
        with th.cuda.stream(side):
            self._launch_all(side.cuda_stream)          # warm-up: one-time function attributes, lazy module load
            side.synchronize()
            H.call("mmd_graph_begin", side.cuda_stream)
            try:
                self._launch_all(side.cuda_stream)
            finally:
                import ctypes
                ex = ctypes.c_void_p()
                H.call("mmd_graph_end", side.cuda_stream, ctypes.byref(ex))
            self.graph = ex
        th.cuda.current_stream(self.device).wait_stream(side)

    def set_step(self, i, shifts=None, noise=None):
        """Refresh the per-step device state: timestep, shifts, noise (video first, then audio - gd:453-454)."""
        self._host_t.fill_(int(i))
        self.t_idx.copy_(self._host_t, non_blocking=True)
        tm = self.tmap[int(i)]
        if self.use_f32:
            self._host_tm.fill_(float(tm) * (1000.0 / self.orig_T))
            self.eng.t_f32.copy_(self._host_tm, non_blocking=True)
        else:
            self._host_tm.fill_(int(tm))
            self.eng.t_i64.copy_(self._host_tm, non_blocking=True)
        self.eng.set_shifts(self.unet.draw_shifts() if shifts is None else shifts)
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
            if self.graph is None:
                # capture replays the step once as warm-up: keep x intact around it
                xv, xa = self.eng.x_video.clone(), self.eng.x_audio.clone()
                self._capture()
                self.eng.x_video.copy_(xv)
                self.eng.x_audio.copy_(xa)
            H.call("mmd_graph_launch", self.graph, H.stream_handle())
        else:
            self.eng.aux.wait_stream(th.cuda.current_stream(self.device))
            self._launch_all(H.stream_handle())

    def step(self, i, shifts=None, noise=None):
        self.set_step(i, shifts, noise)
        self.launch()

    def __del__(self):
        try:
            if self.graph is not None:
                H.lib().mmd_graph_destroy(self.graph)
        except Exception:
            pass
