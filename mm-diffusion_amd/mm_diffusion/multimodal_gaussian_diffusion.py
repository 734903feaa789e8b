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
  * DDIM (`ddim_sample(_loop)`) rides the same graph with mmd_ddim_update; zero-shot conditional sampling
    (`conditional_p_sample_loop`, gd:584-819) is built for both the replacement and the gradient-guided method
    (the latter differentiates the HIP training path w.r.t. the target stream's input)
Not built: cond_fn (the reference's condition_mean / condition_score call `.float()` / `.shape` on the stream
dict, gd:385,403, and raise for every multimodal call), calc_bpd_loop.
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

    def ddim_tables(self, device):
        """[3, T] fp32 alphas_cumprod, alphas_cumprod_prev, alphas_cumprod_next for mmd_ddim_update."""
        key = ("ddim", str(device))
        if key not in self._dev_tables:
            tab = np.stack([self.alphas_cumprod, self.alphas_cumprod_prev, self.alphas_cumprod_next])
            self._dev_tables[key] = th.from_numpy(tab).float().to(device).contiguous()
        return self._dev_tables[key]

    def _tab(self, arr, name, device):
        """fp32 device copy of one fp64 coefficient table (cached by name) for mmd_lincomb_t."""
        key = (name, str(device))
        if key not in self._dev_tables:
            self._dev_tables[key] = th.from_numpy(np.asarray(arr, dtype=np.float64)).float().to(device).contiguous()
        return self._dev_tables[key]

    def _lin(self, a, b, t, ca=None, cb=None, cs=None):
        """(ca[t] a + cb[t] b) cs[t] with per-sample table lookups, one kernel; ca/cb/cs = (name, fp64 table) or None."""
        H.require_cuda(a)
        dev = a.device
        tabs = [None if c is None else self._tab(c[1], c[0], dev) for c in (ca, cb, cs)]
        a32 = a.float().contiguous()
        b32 = None if b is None else b.float().contiguous()
        return ops.lincomb_t(a32, b32, th.empty_like(a32), tabs[0], tabs[1], tabs[2], t.to(th.int64).contiguous())

    def _extract(self, arr, name, t, shape):
        """`_extract_into_tensor` (gd:1289-1303): fp32 table values at t broadcast to `shape`."""
        v = self._tab(arr, name, t.device)[t.long()]
        return v.view(-1, *([1] * (len(shape) - 1))).expand(shape)

    def q_mean_variance(self, x_start, t):
        """gd:170-185."""
        mean = self._lin(x_start, None, t, ca=("sqrt_ac", self.sqrt_alphas_cumprod))
        variance = self._extract(1.0 - self.alphas_cumprod, "one_minus_ac", t, x_start.shape)
        log_variance = self._extract(self.log_one_minus_alphas_cumprod, "log_one_minus_ac", t, x_start.shape)
        return mean, variance, log_variance

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """gd:207-229: mean = coef1 x_0 + coef2 x_t, posterior variance and its clipped log."""
        assert x_start.shape == x_t.shape
        mean = self._lin(x_start, x_t, t, ca=("post_c1", self.posterior_mean_coef1), cb=("post_c2", self.posterior_mean_coef2))
        variance = self._extract(self.posterior_variance, "post_var", t, x_t.shape)
        logvar = self._extract(self.posterior_log_variance_clipped, "post_logvar", t, x_t.shape)
        return mean, variance, logvar

    def _predict_xstart_from_eps(self, x_t, t, eps):
        """gd:345-350."""
        assert x_t.shape == eps.shape
        return self._lin(x_t, eps, t, ca=("sqrt_recip_ac", self.sqrt_recip_alphas_cumprod),
                         cb=("neg_sqrt_recipm1_ac", -self.sqrt_recipm1_alphas_cumprod))

    def _predict_xstart_from_xprev(self, x_t, t, xprev):
        """gd:352-360: (xprev - coef2 x_t) / coef1."""
        assert x_t.shape == xprev.shape
        return self._lin(xprev, x_t, t, ca=("inv_post_c1", 1.0 / self.posterior_mean_coef1),
                         cb=("neg_c2_over_c1", -self.posterior_mean_coef2 / self.posterior_mean_coef1))

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        """gd:362-366: (sqrt_recip_ac x_t - x_0) / sqrt_recipm1_ac."""
        neg1 = ("neg_one", -np.ones_like(self.betas))
        return self._lin(x_t, pred_xstart, t, ca=("sqrt_recip_ac", self.sqrt_recip_alphas_cumprod), cb=neg1,
                         cs=("inv_sqrt_recipm1_ac", 1.0 / self.sqrt_recipm1_alphas_cumprod))

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
    def _update(self, key, model_out, x, t, clip_denoised, noise=None, want=("sample",), denoised_fn=None, start_x=False):
        """Run the fused update kernel for one stream; returns a dict with the requested outputs.  denoised_fn (gd:263-268): the x_0
        prediction goes through it BEFORE the clamp - one pass of the kernel for the unclamped prediction, the caller's function on that
        tensor, then the kernel again with the processed tensor in the place of the model's mean channels, read as an x_0 prediction."""
        if denoised_fn is not None:
            raw = self._update(key, model_out, x, t, False, want=("pred_xstart",), start_x=start_x)["pred_xstart"]
            x0 = denoised_fn(raw)
            cd = 2 if x.dim() == 5 else 1
            mo2 = model_out.float().clone()
            mo2.narrow(cd, 0, x.shape[cd]).copy_(x0)
            return self._update(key, mo2, x, t, clip_denoised, noise=noise, want=want, start_x=True)
        tab, _ = self.device_tables(x.device)
        F, C, HW = _geom(x)
        xs = x.float().contiguous()
        mo = model_out.float().contiguous()
        res = {k: th.empty_like(xs) for k in want}
        ops.ddpm_update(xs, mo, noise, res.get("sample"), tab, t.to(th.int64).contiguous(), F, C, HW,
                        self._flags(clip_denoised) | (2 if start_x else 0), x0_out=res.get("pred_xstart"), mean_out=res.get("mean"),
                        logvar_out=res.get("log_variance"))
        return res

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gd:231-343.  Returns the same nested dict ({'mean','variance','log_variance','pred_xstart','model_predict'}
        each {'video','audio'})."""
        model_kwargs = model_kwargs or {}
        B = x["video"].shape[0]
        assert t.shape == (B,)
        video_output, audio_output = model(x["video"], x["audio"], self._scale_timesteps(t), **model_kwargs)
        out = {k: {} for k in ("mean", "variance", "log_variance", "pred_xstart", "model_predict")}
        for key, mo in (("video", video_output), ("audio", audio_output)):
            r = self._update(key, mo, x[key], t, clip_denoised, want=("mean", "log_variance", "pred_xstart"), denoised_fn=denoised_fn)
            out["mean"][key], out["log_variance"][key], out["pred_xstart"][key] = r["mean"], r["log_variance"], r["pred_xstart"]
            out["variance"][key] = th.exp(r["log_variance"])
            out["model_predict"][key] = mo
        return out

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """One ancestral step (gd:415-474).  Like the reference, the `noise` argument is ignored and fresh N(0,1)
        noise is drawn for both streams (video first), also at t == 0."""
        if cond_fn is not None:
            raise NotImplementedError("cond_fn: the reference's condition_mean raises for the multimodal dict (gd:385 calls dict.float()); not built")
        model_kwargs = model_kwargs or {}
        video_output, audio_output = model(x["video"], x["audio"], self._scale_timesteps(t), **model_kwargs)
        noise = {"video": self._randn_like(x["video"]), "audio": self._randn_like(x["audio"])}
        res = {"sample": {}, "pred_start": {}, "pred_noise": {"video": video_output, "audio": audio_output}}
        differentiable = th.is_grad_enabled() and any(v.requires_grad for v in (video_output, audio_output, x["video"], x["audio"]))
        for key, mo in (("video", video_output), ("audio", audio_output)):
            if differentiable:          # gradient-guided conditional sampling differentiates the step (gd:795-817)
                from .train_ops import DdpmUpdateFn
                tab, _ = self.device_tables(x[key].device)
                s_, x0_ = DdpmUpdateFn.apply(x[key].float(), mo.float(), noise[key].float().contiguous(), tab,
                                             t.to(th.int64).contiguous(), self._flags(clip_denoised), _geom(x[key]))
                res["sample"][key], res["pred_start"][key] = s_, x0_
                continue
            r = self._update(key, mo, x[key], t, clip_denoised, noise=noise[key].float().contiguous(),
                             want=("sample", "pred_xstart"), denoised_fn=denoised_fn)
            res["sample"][key], res["pred_start"][key] = r["sample"], r["pred_xstart"]
        return res

    # ------------------------------------------------------------------ DDIM
    def _ddim(self, model, x, t, clip_denoised, model_kwargs, eta, reverse):
        video_output, audio_output = model(x["video"], x["audio"], self._scale_timesteps(t), **(model_kwargs or {}))
        res = {"sample": {}, "pred_xstart": {}}
        flags = self._flags(clip_denoised) | (8 if reverse else 0)
        for key, mo in (("video", video_output), ("audio", audio_output)):
            tab, _ = self.device_tables(x[key].device)
            F, C, HW = _geom(x[key])
            xs = x[key].float().contiguous()
            noise = None if reverse else self._randn_like(xs).float().contiguous()      # drawn even when eta == 0 (gd:876-877)
            out, x0 = th.empty_like(xs), th.empty_like(xs)
            ops.ddim_update(xs, mo.float().contiguous(), noise, out, tab, self.ddim_tables(xs.device), t.to(th.int64).contiguous(),
                            F, C, HW, flags, eta, x0_out=x0)
            res["sample"][key], res["pred_xstart"][key] = out, x0
        return res

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """One DDIM step (gd:821-901): eps re-derived from the (clipped) x_0 prediction; noise drawn video first."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn: the reference's condition_score raises for the multimodal dict (gd:403)")
        return self._ddim(model, x, t, clip_denoised, model_kwargs, eta, False)

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """DDIM reverse ODE step x_t -> x_{t+1} (gd:903-953).  The reference indexes its own result dict the wrong way
        round (`out["video"]["pred_xstart"]`, gd:925) and raises KeyError; this computes the intended update."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not supported by the fused update kernel")
        return self._ddim(model, x, t, clip_denoised, model_kwargs, 0.0, True)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=True, eta=0.0):
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                        cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                                                        eta=eta):
            final = sample
        return final

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, use_graph=True):
        """gd:989-1046 (the `noise` argument is ignored there too: x_T is always drawn on the CPU, video then audio)."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn: see ddim_sample")
        device = self._sampling_device(device)
        x = {"video": th.randn(*shape["video"], device="cpu").to(device), "audio": th.randn(*shape["audio"], device="cpu").to(device)}
        indices = self._indices(progress)
        from .sampler import GraphStepper, unwrap_unet
        unet = unwrap_unet(model)
        if use_graph and unet is not None and not (model_kwargs or {}):
            stepper = GraphStepper(self, unet, shape["video"][0], device, clip_denoised, update="ddim", eta=eta)
            stepper.load(x["video"], x["audio"])
            try:
                for i in indices:
                    stepper.step(i)
                    yield stepper.current()
            finally:
                stepper.close()
            return
        for i in indices:
            t = th.tensor([i] * shape["video"][0], device=device)
            with th.no_grad():
                out = self.ddim_sample(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs, eta=eta)
            yield out["sample"]
            x = out["sample"]

    # ------------------------------------------------------------------ zero-shot conditional sampling (gd:584-819)
    def _sampling_device(self, device):
        if device is None:
            from . import dist_util
            device = dist_util.dev()
        if th.device(device).type != "cuda":
            raise H.MMDError("sampling runs on the MI355X HIP path only (device must be a GPU); no CPU fallback")
        return device

    def _indices(self, progress):
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        return indices

    def conditional_p_sample_loop(self, model, shape, use_fp16, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=True, class_scale=0.):
        """class_scale == 0: replacement method; otherwise the gradient-guided method (gd:584-640)."""
        fn = self.conditional_p_sample_loop_progressive_unscale if class_scale == 0 else self.conditional_p_sample_loop_progressive_scale
        final = None
        for sample in fn(model, shape, use_fp16, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                         model_kwargs=model_kwargs, device=device, progress=progress, class_scale=class_scale):
            final = sample
        return final

    def _cond_setup(self, shape, noise, model_kwargs, device):
        device = self._sampling_device(device)
        if noise is None:
            noise = {"video": th.randn(*shape["video"], device="cpu").to(device), "audio": th.randn(*shape["audio"], device="cpu").to(device)}
        model_kwargs = model_kwargs if model_kwargs is not None else {}
        cond = {k: model_kwargs.pop(k) for k in ("video", "audio") if k in model_kwargs}     # popped like gd:687-691
        return device, noise, model_kwargs, cond

    def conditional_p_sample_loop_progressive_unscale(self, model, shape, use_fp16, noise=None, clip_denoised=True, denoised_fn=None,
                                                      cond_fn=None, model_kwargs=None, device=None, progress=False, class_scale=0.0,
                                                      use_graph=True):
        """Replacement method (gd:642-720): before every step the conditioning stream is overwritten with
        q_sample(condition, t, noise=<the fixed initial noise of that stream>); then one ordinary p_sample."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn: see p_sample")
        device, noise, model_kwargs, cond = self._cond_setup(shape, noise, model_kwargs, device)
        x = dict(noise)
        B = shape["video"][0]
        from .sampler import GraphStepper, unwrap_unet
        unet = unwrap_unet(model)
        stepper = None
        if use_graph and unet is not None and not model_kwargs:
            stepper = GraphStepper(self, unet, B, device, clip_denoised)
            stepper.load(x["video"], x["audio"])
        for i in self._indices(progress):
            t = th.tensor([i] * B, device=device)
            for k in ("video", "audio"):
                if k in cond:
                    x[k] = self.q_sample(cond[k], t, noise=noise[k])
                    if stepper is not None:
                        stepper.set_x(k, x[k])
            if stepper is not None:
                stepper.step(i)
                x = stepper.current()
            else:
                with th.no_grad():
                    x = self.p_sample(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)["sample"]
            yield x

    def conditional_p_sample_loop_progressive_scale(self, model, shape, use_fp16, noise=None, clip_denoised=True, denoised_fn=None,
                                                    cond_fn=None, model_kwargs=None, device=None, progress=False, class_scale=3.0):
        """Gradient-guided method (gd:722-817).  Per step: replace the conditioning stream at t, take one differentiable
        p_sample, loss = mean_flat((sample[cond] - q_sample(cond, t-1))^2).mean() (x 2^20 when use_fp16, never unscaled -
        as in the reference), and move the target stream against d loss / d x_t[target] scaled by
        class_scale * sqrt_alphas_cumprod[i].  The U-Net backward w.r.t. its input runs on the HIP training kernels."""
        if cond_fn is not None or denoised_fn is not None:
            raise NotImplementedError("cond_fn / denoised_fn: see p_sample")
        device, noise, model_kwargs, cond = self._cond_setup(shape, noise, model_kwargs, device)
        if not cond:
            raise UnboundLocalError("conditional sampling needs model_kwargs['video'] or model_kwargs['audio'] (gd:776-786)")
        x = dict(noise)
        B = shape["video"][0]
        from .sampler import unwrap_unet
        unet = unwrap_unet(model)
        frozen = [p for p in unet.parameters() if p.requires_grad] if unet is not None else []
        for p in frozen:                 # only d/dx is needed: skip every weight-gradient kernel
            p.requires_grad_(False)
        try:
            yield from self._guided_steps(model, x, noise, cond, B, device, use_fp16, clip_denoised, model_kwargs, progress, class_scale)
        finally:
            for p in frozen:
                p.requires_grad_(True)

    def _guided_steps(self, model, x, noise, cond, B, device, use_fp16, clip_denoised, model_kwargs, progress, class_scale):
        for i in self._indices(progress):
            t = th.tensor([i] * B, device=device)
            for k, other in (("video", "audio"), ("audio", "video")):
                if k in cond:
                    condition, target = k, other
                    x[condition] = self.q_sample(cond[k], t, noise=noise[condition])
                    # at i == 0 the reference indexes its tables with t - 1 == -1, i.e. the LAST entry (gd:779,785)
                    previous_step_condition = self.q_sample(cond[k], (t - 1) % self.num_timesteps, noise=noise[condition])
            with th.enable_grad():
                none_zero_mask = (t != 0).float().view(-1, *([1] * (len(x[target].shape) - 1)))
                x[target] = x[target].detach().requires_grad_()
                out = self.p_sample(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
                pred = out["sample"]
                loss = mean_flat((pred[condition] - previous_step_condition) ** 2)
                loss_scale = 2 ** 20 if use_fp16 == True else 1.       # noqa: E712 (reference compares with ==)
                grad = th.autograd.grad(loss.mean() * loss_scale, x[target])[0]
                x = {condition: x[condition], target: (pred[target] - none_zero_mask * grad * class_scale *
                                                       float(self.sqrt_alphas_cumprod[i])).detach()}
            yield x

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
        every step then replays one captured hipGraph (U-Net plan + both fused updates).  With a denoised_fn (a host callable on the x_0
        prediction, gd:263-268) the steps are launched eagerly through p_sample."""
        if cond_fn is not None:
            raise NotImplementedError("cond_fn: see p_sample")
        device = self._sampling_device(device)
        video = th.randn(*shape["video"], device="cpu").to(device)
        audio = th.randn(*shape["audio"], device="cpu").to(device)
        x = {"video": video, "audio": audio}
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        from .sampler import GraphStepper, unwrap_unet
        unet = unwrap_unet(model)
        if use_graph and unet is not None and not (model_kwargs or {}) and denoised_fn is None:
            stepper = GraphStepper(self, unet, shape["video"][0], device, clip_denoised)
            stepper.load(x["video"], x["audio"])
            try:
                for i in indices:
                    stepper.step(i)
                    yield stepper.current()
            finally:
                stepper.close()
            return
        for i in indices:
            t = th.tensor([i] * shape["video"][0], device=device)
            with th.no_grad():
                out = self.p_sample(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
            yield out["sample"]
            x = out["sample"]

    # ------------------------------------------------------------------ training loss (forward value only for now)
    def multimodal_training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """gd:1114-1203: per-sample loss = mse_video + mse_audio (+ vb_video + vb_audio with learned-range variance; the vb term
        uses the frozen mean and clip_denoised=False, gd:1147-1174).  q_sample, the U-Net forward and the loss reductions run in
        libmmd; with autograd enabled the terms are differentiable (mmd_mse_grad / mmd_loss_terms_bwd feed the U-Net backward)."""
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
            # differentiable loss: per-sample terms with their gradient kernels
            if learned:
                from .train_ops import LossTermsFn
                for key, mo in (("video", video_output), ("audio", audio_output)):
                    term[f"mse_{key}"], term[f"vb_{key}"] = LossTermsFn.apply(
                        mo, tgt[key].float().contiguous(), x_start[key].float().contiguous(), xt[key], tab, t64, _geom(xt[key]), flags, vb_scale)
                term["loss"] = (term["vb_video"] + term["vb_audio"]) + (term["mse_video"] + term["mse_audio"])
                return term
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
