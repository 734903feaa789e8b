"""Multimodal Gaussian diffusion (DDPM ancestral sampling / training loss) on the MI355X HIP path.

Same public surface as the reference's `GaussianDiffusion`
(/root/reference/mm_diffusion/multimodal_gaussian_diffusion.py:102-1203): enums, `get_named_beta_schedule`,
fp64 numpy tables (`betas`, `alphas_cumprod`, ... same attribute names), `q_sample`, `p_mean_variance`,
`p_sample`, `p_sample_loop(_progressive)`, `multimodal_training_losses`.

MI355X-first differences:
  * all per-step coefficient tables are uploaded ONCE as a [7, T] fp32 device table; the reference re-uploads
    ~16 fp64 tables per step (`_extract_into_tensor`, gd:1289-1303)
  * the whole epsilon -> x0 -> clamp -> posterior mean -> sample chain of p_mean_variance + p_sample is one
    fused elementwise kernel per stream (mmd_ddpm_update)
  * `p_sample_loop` captures [U-Net launch plan + both updates] into one hipGraph and replays it per step,
    refreshing only the timestep, the window shifts and the noise
Not built (out of the hot path): DDIM loops, zero-shot conditional sampling, bpd loops.
"""
import enum
import math

import numpy as np
import torch as th

from . import _hip as H
from . import ops


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """Named beta schedules (reference gd:17-41)."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    T = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta) for i in range(T)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def _geom(x):
    """API tensor -> (F, C, HW) of the [N, F, C, HW] view the update kernels index (audio: F = 1)."""
    if x.dim() == 5:
        return x.shape[1], x.shape[2], x.shape[3] * x.shape[4]
    if x.dim() == 3:
        return 1, x.shape[1], x.shape[2]
    raise ValueError(f"expected video [N,F,C,H,W] or audio [N,C,L], got {tuple(x.shape)}")


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        if model_mean_type == ModelMeanType.PREVIOUS_X:
            raise NotImplementedError("ModelMeanType.PREVIOUS_X is never produced by the factory (msu:225-242)")
        if model_var_type == ModelVarType.LEARNED:
            raise NotImplementedError("ModelVarType.LEARNED is never produced by the factory (msu:225-242)")

        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])

        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        # noise source for sampling: callable(like_tensor) -> N(0,1) tensor; default th.randn_like (gd:453-454)
        self.noise_source = None
        self._dev_tables = {}

    # ------------------------------------------------------------------ device tables
    def _fixed_logvar(self):
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            return self.posterior_log_variance_clipped
        return np.log(np.append(self.posterior_variance[1], self.betas[1:]))     # FIXED_LARGE (gd:288-291)

    def device_tables(self, device):
        """[7, T] fp32: sqrt_recip_ac, sqrt_recipm1_ac, post_c1, post_c2, fixed logvar, min_log, max_log; and [2, T]
        q_sample table.  fp64 -> fp32 exactly as the reference's `.float()` after indexing (gd:1300)."""
        key = str(device)
        if key not in self._dev_tables:
            tab = np.stack([self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
                            self.posterior_mean_coef2, self._fixed_logvar(), self.posterior_log_variance_clipped,
                            np.log(self.betas)])
            q = np.stack([self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod])
            self._dev_tables[key] = (th.from_numpy(tab).float().to(device).contiguous(),
                                     th.from_numpy(q).float().to(device).contiguous())
        return self._dev_tables[key]

    def _flags(self, clip_denoised):
        return ((1 if clip_denoised else 0) | (2 if self.model_mean_type == ModelMeanType.START_X else 0) |
                (4 if self.model_var_type == ModelVarType.LEARNED_RANGE else 0))

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _randn_like(self, x):
        return self.noise_source(x) if self.noise_source is not None else th.randn_like(x)

    # ------------------------------------------------------------------ q(x_t | x_0)
    def q_sample(self, x_start, t, noise=None):
        """x_t = sqrt(ac_t) x_0 + sqrt(1-ac_t) eps (gd:187-205) as one kernel."""
        if noise is None:
            noise = self._randn_like(x_start)
        assert noise.shape == x_start.shape
        H.require_cuda(x_start)
        _, qtab = self.device_tables(x_start.device)
        out = th.empty_like(x_start, dtype=th.float32)
        ops.q_sample(x_start.float().contiguous(), noise.float().contiguous(), out, qtab, t.to(th.int64).contiguous())
        return out

    # ------------------------------------------------------------------ p(x_{t-1} | x_t)
    def _update(self, key, model_out, x, t, clip_denoised, noise=None, want=("sample",)):
        """Run the fused update kernel for one stream; returns a dict with the requested outputs."""
        tab, _ = self.device_tables(x.device)
        F, C, HW = _geom(x)
        xs = x.float().contiguous()
        mo = model_out.float().contiguous()
        res = {k: th.empty_like(xs) for k in want}
        ops.ddpm_update(xs, mo, noise, res.get("sample"), tab, t.to(th.int64).contiguous(), F, C, HW,
                        self._flags(clip_denoised), x0_out=res.get("pred_xstart"), mean_out=res.get("mean"),
                        logvar_out=res.get("log_variance"))
        return res

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gd:231-343.  Returns the same nested dict ({'mean','variance','log_variance','pred_xstart','model_predict'}
        each {'video','audio'})."""
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not supported by the fused update kernel")
        model_kwargs = model_kwargs or {}
        B = x["video"].shape[0]
        assert t.shape == (B,)
        video_output, audio_output = model(x["video"], x["audio"], self._scale_timesteps(t), **model_kwargs)
        out = {k: {} for k in ("mean", "variance", "log_variance", "pred_xstart", "model_predict")}
        for key, mo in (("video", video_output), ("audio", audio_output)):
            r = self._update(key, mo, x[key], t, clip_denoised, want=("mean", "log_variance", "pred_xstart"))
            out["mean"][key], out["log_variance"][key], out["pred_xstart"][key] = r["mean"], r["log_variance"], r["pred_xstart"]
            out["variance"][key] = th.exp(r["log_variance"])
            out["model_predict"][key] = mo
        return out

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """One ancestral step (gd:415-474).  Like the reference, the `noise` argument is ignored and fresh N(0,1)
        noise is drawn for both streams (video first), also at t == 0."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn (zero-shot conditional sampling) are not built (SURVEY 8f4)")
        model_kwargs = model_kwargs or {}
        video_output, audio_output = model(x["video"], x["audio"], self._scale_timesteps(t), **model_kwargs)
        noise = {"video": self._randn_like(x["video"]), "audio": self._randn_like(x["audio"])}
        res = {"sample": {}, "pred_start": {}, "pred_noise": {"video": video_output, "audio": audio_output}}
        for key, mo in (("video", video_output), ("audio", audio_output)):
            r = self._update(key, mo, x[key], t, clip_denoised, noise=noise[key].float().contiguous(),
                             want=("sample", "pred_xstart"))
            res["sample"][key], res["pred_start"][key] = r["sample"], r["pred_xstart"]
        return res

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=True):
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                     device=device, progress=progress):
            final = sample
        return final

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, use_graph=True):
        """gd:523-582.  x_T is drawn on the CPU (video, then audio) and moved to the device like the reference;
        every step then replays one captured hipGraph (U-Net plan + both fused updates)."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn are not built (SURVEY 8f4)")
        if device is None:
            from . import dist_util
            device = dist_util.dev()
        if th.device(device).type != "cuda":
            raise H.MMDError("sampling runs on the MI355X HIP path only (device must be a GPU); no CPU fallback")
        video = th.randn(*shape["video"], device="cpu").to(device)
        audio = th.randn(*shape["audio"], device="cpu").to(device)
        x = {"video": video, "audio": audio}
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        from .sampler import GraphStepper, unwrap_unet
        unet = unwrap_unet(model)
        if use_graph and unet is not None and not (model_kwargs or {}):
            stepper = GraphStepper(self, unet, shape["video"][0], device, clip_denoised)
            stepper.load(x["video"], x["audio"])
            for i in indices:
                stepper.step(i)
                yield stepper.current()
            return
        for i in indices:
            t = th.tensor([i] * shape["video"][0], device=device)
            with th.no_grad():
                out = self.p_sample(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
            yield out["sample"]
            x = out["sample"]

    # ------------------------------------------------------------------ training loss (forward value only for now)
    def multimodal_training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """gd:1114-1203: per-sample loss = mse_video + mse_audio (+ vb_video + vb_audio with learned-range variance; the
        vb term uses the frozen mean and clip_denoised=False, gd:1147-1174).  Forward VALUES only: q_sample, the U-Net
        forward and the loss reductions all run in libmmd; the backward pass is not built yet (cfg4, next)."""
        model_kwargs = model_kwargs or {}
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError("KL / RESCALED_KL losses produce no terms in the reference either (gd:1143)")
        if noise is None:
            noise = {"video": self._randn_like(x_start["video"]), "audio": self._randn_like(x_start["audio"])}
        xt = {k: self.q_sample(x_start[k], t, noise=noise[k]) for k in ("video", "audio")}
        video_output, audio_output = model(xt["video"], xt["audio"], self._scale_timesteps(t), **model_kwargs)
        tab, _ = self.device_tables(xt["video"].device)
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE
        flags = (2 if self.model_mean_type == ModelMeanType.START_X else 0) | (4 if learned else 0)
        vb_scale = self.num_timesteps / 1000.0 if self.loss_type == LossType.RESCALED_MSE else 1.0
        tgt = x_start if self.model_mean_type == ModelMeanType.START_X else noise
        t64 = t.to(th.int64).contiguous()
        term = {}
        if th.is_grad_enabled() and (video_output.requires_grad or audio_output.requires_grad):
            # differentiable loss: per-sample mse with its gradient kernel (the vb term's backward is not built yet)
            if learned:
                raise NotImplementedError("gradients of the learned-sigma vb term are not built yet; train with learn_sigma=False")
            from .train_ops import MseLossFn
            term["mse_video"] = MseLossFn.apply(video_output, tgt["video"])
            term["mse_audio"] = MseLossFn.apply(audio_output, tgt["audio"])
            term["loss"] = term["mse_video"] + term["mse_audio"]
            return term
        for key, mo in (("video", video_output), ("audio", audio_output)):
            F, C, HW = _geom(xt[key])
            mse, vb = ops.loss_terms(mo.float().contiguous(), tgt[key].float().contiguous(), tab, t64, F, C, HW, flags,
                                     x0=x_start[key].float().contiguous() if learned else None,
                                     xt=xt[key] if learned else None, vb_scale=vb_scale)
            if learned:
                term[f"vb_{key}"] = vb
            term[f"mse_{key}"] = mse
        # same accumulation order as the reference: (vb_video + vb_audio) then (mse_video + mse_audio)
        term["loss"] = (term["vb_video"] + term["vb_audio"] if learned else 0) + (term["mse_video"] + term["mse_audio"])
        return term
