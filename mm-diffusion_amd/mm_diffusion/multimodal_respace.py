"""Timestep respacing (reference /root/reference/mm_diffusion/multimodal_respace.py:1-139): same public names -
`space_timesteps`, `SpacedDiffusion`, `_WrappedModel` - same tables and timestep_map."""
import numpy as np
import torch as th

from .multimodal_gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Set of original timesteps to keep.  "ddimN" = fixed integer stride giving exactly N steps; otherwise a
    comma list / list of per-section counts over equal sections (resp:6-59)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, keep = 0, []
    for i, cnt in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            keep.append(start + round(cur))
            cur += stride
        start += size
    return set(keep)


class SpacedDiffusion(GaussianDiffusion):
    """Diffusion over a retained subset of the base process' timesteps (resp:62-125)."""

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

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def p_sample(self, model, *args, **kwargs):
        return super().p_sample(self._wrap_model(model), *args, **kwargs)

    def _ddim(self, model, *args, **kwargs):          # ddim_sample / ddim_reverse_sample reach the model through here
        return super()._ddim(self._wrap_model(model), *args, **kwargs)

    def multimodal_training_losses(self, model, *args, **kwargs):
        return super().multimodal_training_losses(self._wrap_model(model), *args, **kwargs)

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
    """Maps the loop index to the original timestep before calling the model (resp:127-139).  The map lives on
    the device once per device instead of being rebuilt from a Python list every call."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = {}

    def __call__(self, video_x, audio_x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        if key not in self._maps:
            self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = self._maps[key][ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(video_x, audio_x, new_ts, **kwargs)
