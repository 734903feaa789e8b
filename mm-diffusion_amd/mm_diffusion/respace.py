"""Timestep respacing for the tensor-valued (SR stage) diffusion (reference mm_diffusion/respace.py:7-130): the same
`space_timesteps` grammar (incl. "ddimN") and beta re-derivation as multimodal_respace, wrapping a single-tensor model."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion
from .multimodal_respace import space_timesteps  # noqa: F401


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base_ac = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64))
        last, new_betas = 1.0, []
        for i, ac in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _model_out(self, model, x, t, model_kwargs):
        return super()._model_out(self._wrap_model(model), x, t, model_kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        if not hasattr(self, "_map_cache"):
            self._map_cache = {}          # device copies of timestep_map, shared by every wrapper (graph capture forbids new H2D copies)
        w = _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)
        w._maps = self._map_cache
        return w

    def _scale_timesteps(self, t):
        return t          # scaling is done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model, self.timestep_map = model, timestep_map
        self.rescale_timesteps, self.original_num_steps = rescale_timesteps, original_num_steps
        self._maps = {}

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        if key not in self._maps:
            self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = self._maps[key][ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
