"""Tensor-valued Gaussian diffusion for the image super-resolution stage (reference mm_diffusion/gaussian_diffusion.py:
119-795; driven by multimodal_sample_sr.py:186-253) on the MI355X HIP path.

Shares the schedule tables, the fused update kernels (mmd_ddpm_update / mmd_ddim_update, learned-range variance included)
and the helper combinations with the multimodal class; only the sampling surface differs: states are `[N, C, H, W]` tensors,
`p_sample_loop` / `ddim_sample_loop` take `noise=` and `model_kwargs=` (the SR script passes `low_res` and a noise tensor
repeated over the frames of a clip) and return the final sample tensor.

Reference behaviours kept: `p_sample_loop` hands its `noise` argument down to `p_sample`, so when a start noise is given
the SAME tensor is re-used as the per-step noise of every step (gd:547-556, 423-424); `ddim_sample` always draws fresh noise
(gd:661); `ddim_sample_loop_progressive` defaults to eta = 0.5 while `ddim_sample_loop` defaults to 0.0 (gd:725,759).
Not built: cond_fn (classifier guidance), training_losses / bpd loops of the SR model (training the SR stage is out of the
hot path)."""
import torch as th

from . import _hip as H
from . import ops
from .multimodal_gaussian_diffusion import (GaussianDiffusion as _Base, LossType, ModelMeanType, ModelVarType,  # noqa: F401
                                            get_named_beta_schedule, betas_for_alpha_bar, mean_flat)


def _geom4(x):
    if x.dim() == 4:
        return 1, x.shape[1], x.shape[2] * x.shape[3]
    raise ValueError(f"expected an image batch [N,C,H,W], got {tuple(x.shape)}")


class GaussianDiffusion(_Base):
    """Same constructor / tables as the multimodal class (gd:119-170)."""

    def _model_out(self, model, x, t, model_kwargs):
        out = model(x, self._scale_timesteps(t), **(model_kwargs or {}))
        return out.float().contiguous()

    def _upd(self, model_output, x, t, clip_denoised, noise, want):
        tab, _ = self.device_tables(x.device)
        F, C, HW = _geom4(x)
        xs = x.float().contiguous()
        res = {k: th.empty_like(xs) for k in want}
        ops.ddpm_update(xs, model_output, noise, res.get("sample"), tab, t.to(th.int64).contiguous(), F, C, HW, self._flags(clip_denoised),
                        x0_out=res.get("pred_xstart"), mean_out=res.get("mean"), logvar_out=res.get("log_variance"))
        return res

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gd:230-328 -> {'mean', 'variance', 'log_variance', 'pred_xstart'} tensors."""
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not supported by the fused update kernel")
        H.require_cuda(x)
        assert t.shape == (x.shape[0],)
        mo = self._model_out(model, x, t, model_kwargs)
        r = self._upd(mo, x, t, clip_denoised, None, ("mean", "log_variance", "pred_xstart"))
        r["variance"] = th.exp(r["log_variance"])
        return r

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """gd:400-449: fresh N(0,1) noise unless a noise tensor is handed in."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn are not built for the SR stage")
        H.require_cuda(x)
        mo = self._model_out(model, x, t, model_kwargs)
        if noise is None:
            noise = self._randn_like(x)
        r = self._upd(mo, x, t, clip_denoised, noise.float().contiguous(), ("sample", "pred_xstart"))
        return {"sample": r["sample"], "pred_xstart": r["pred_xstart"]}

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                      progress=True, cond_range=[0, 1000]):
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                     cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                                                     cond_range=cond_range):
            final = sample
        return final["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                                  device=None, progress=False, cond_range=[0, 1000]):
        if cond_fn is not None:
            raise NotImplementedError("cond_fn is not built for the SR stage")
        device = self._sampling_device(device)
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device="cpu").to(device)
        for i in self._indices(progress):
            t = th.tensor([i] * shape[0], device=device)
            with th.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs, noise=noise)
            yield out
            img = out["sample"]

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """gd:626-674."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn are not built for the SR stage")
        H.require_cuda(x)
        mo = self._model_out(model, x, t, model_kwargs)
        tab, _ = self.device_tables(x.device)
        F, C, HW = _geom4(x)
        xs = x.float().contiguous()
        noise = self._randn_like(xs).float().contiguous()
        out, x0 = th.empty_like(xs), th.empty_like(xs)
        ops.ddim_update(xs, mo, noise, out, tab, self.ddim_tables(xs.device), t.to(th.int64).contiguous(), F, C, HW, self._flags(clip_denoised),
                        eta, x0_out=x0)
        return {"sample": out, "pred_xstart": x0}

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                         progress=False, eta=0.0):
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                        cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress, eta=eta):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                                     device=None, progress=False, eta=0.5):
        device = self._sampling_device(device if device is not None else next(model.parameters()).device)
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        for i in self._indices(progress):
            t = th.tensor([i] * shape[0], device=device)
            with th.no_grad():
                out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta)
            yield out
            img = out["sample"]

    # the multimodal dict-valued entry points do not apply to the tensor-valued process
    def multimodal_training_losses(self, *a, **kw):
        raise NotImplementedError("tensor-valued diffusion: use the multimodal class for {'video','audio'} states")

    conditional_p_sample_loop = multimodal_training_losses
