"""Tensor-valued DPM-Solver / DPM-Solver++ for the super-resolution stage (reference mm_diffusion/dpm_solver_plus.py, used by
multimodal_sample_sr.py:199-228 as `singlemodal_DPM_Solver(model=sr_model, alphas_cumprod=..., predict_x0=..., model_kwargs=
{"low_res": frames}).sample(noise, steps=50, order=2, skip_type="time_uniform", method="multistep")`).

The solver arithmetic is the multimodal driver's (multimodal_dpm_solver_plus.DPM_Solver: host fp32 coefficients, one fused
lincomb kernel per update) run on a single stream: the state tensor is carried as {"x": tensor}, so `sample` accepts and returns
plain tensors and the update methods take / return {"x": tensor} dicts.  Differences from the multimodal class that follow the
reference: the model is called as `model(x, t_input, **model_kwargs)` and a 6-channel (learned sigma) output is cut to its first
3 channels (dpm_solver_plus.py:299-304); there is no audio stream, hence no x0-form coefficient quirk in the first-order update.
"""
import torch

from .multimodal_dpm_solver_plus import DPM_Solver as _MultimodalSolver
from .multimodal_dpm_solver_plus import NoiseScheduleVP, interpolate_fn, expand_dims  # noqa: F401


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", rescale=False, **unused):
    if model_type != "noise" or guidance_type != "uncond":
        raise NotImplementedError("only the unguided noise-prediction wrapper is built (what multimodal_sample_sr.py uses)")

    def get_model_input_time(t_continuous):
        if noise_schedule.schedule == "discrete":
            max_step = 1000. if rescale else noise_schedule.total_N
            return ((t_continuous - 1. / noise_schedule.total_N) * max_step).to(torch.int)
        return t_continuous

    def model_fn(x, t_continuous):
        xt = x["x"]
        if t_continuous.reshape((-1,)).shape[0] == 1:
            t_continuous = t_continuous.expand((xt.shape[0]))
        output = model(xt, get_model_input_time(t_continuous), **model_kwargs)
        if getattr(model, "out_channels", None) == 6:
            output = output[:, :3, ...]
        return {"x": output}

    return model_fn


class DPM_Solver(_MultimodalSolver):
    KEYS = ("x",)

    def __init__(self, model, betas=None, alphas_cumprod=None, predict_x0=False, thresholding=False, max_val=1., model_kwargs={}, rescale=False):
        noise_schedule = NoiseScheduleVP(schedule="discrete", betas=betas, alphas_cumprod=alphas_cumprod)
        self.model = model_wrapper(model, noise_schedule, model_type="noise", model_kwargs=model_kwargs, rescale=rescale)
        self.noise_schedule = noise_schedule
        self.predict_x0, self.thresholding, self.max_val, self.rescale = predict_x0, thresholding, max_val, rescale
        self.nfe = 0

    def sample(self, x, *args, **kwargs):
        out = super().sample({"x": x} if torch.is_tensor(x) else x, *args, **kwargs)
        return out["x"] if torch.is_tensor(x) else out
