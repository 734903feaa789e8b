#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the reference itself
never travels to the GPU box - only the small .npz/.json data files written
here do.  Re-run with:  python tools/gen_golden.py [--only NAME ...]

Protocol (see tests/golden/README.md):
  * weights : mm_diffusion.synth.synth_tensor(key, shape)  (key-seeded, no files)
  * shifts  : python `random.seed(SEED)`; the reference draws
              random.randint(0, F-window) once per shifted CrossAttentionBlock
              call (multimodal_unet.py:619-620); we also RECORD the sequence
  * noise   : torch.manual_seed(SEED) on CPU, draw order as the reference
              (multimodal_gaussian_diffusion.py:547-550, 453-454)
"""
import argparse
import json
import os
import random
import sys
import types

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
from mm_diffusion.synth import synth_tensor  # noqa: E402  (our own key-seeded init)

# ---- import the reference under the name `ref_mm` (blobfile / mpi4py are absent) ----
for _m in ("blobfile", "mpi4py"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["mpi4py"].MPI = None
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ref_mm", "/root/reference/mm_diffusion/__init__.py",
    submodule_search_locations=["/root/reference/mm_diffusion"])
ref_mm = importlib.util.module_from_spec(_spec)
sys.modules["ref_mm"] = ref_mm
_spec.loader.exec_module(ref_mm)
from ref_mm import multimodal_script_util as msu  # noqa: E402
from ref_mm import multimodal_unet as runet  # noqa: E402
from ref_mm import multimodal_gaussian_diffusion as rgd  # noqa: E402
from ref_mm import multimodal_respace as rresp  # noqa: E402
from ref_mm import nn as rnn  # noqa: E402

th.set_num_threads(8)

CONFIGS = {
    # F=8, 16x16 video, 512-sample audio: every level down to a 2x2 frame / 1 audio token per frame
    "tiny": dict(video_size=[8, 3, 16, 16], audio_size=[1, 512], num_channels=64,
                 num_head_channels=32, num_res_blocks=1, channel_mult="1,2,3,4",
                 resblock_updown=True),
    # F=16, 32x32, 6400 samples: audio 100 tokens at ds=8 -> L % F != 0 (remainder path)
    "mid": dict(video_size=[16, 3, 32, 32], audio_size=[1, 6400], num_channels=64,
                num_head_channels=32, num_res_blocks=1, channel_mult="1,2,3,4",
                resblock_updown=True),
    # shipped Landscape / AIST++ base model (ssh_scripts/multimodal_sample_sr.sh:3-8)
    "full": dict(video_size=[16, 3, 64, 64], audio_size=[1, 25600], num_channels=128,
                 num_head_channels=64, num_res_blocks=2, resblock_updown=True),
}


def flags(name, **over):
    d = msu.model_and_diffusion_defaults()
    d.update(CONFIGS[name])
    d.update(over)
    return d


def synth_init(model):
    sd = model.state_dict()
    model.load_state_dict({k: synth_tensor(k, v.shape) for k, v in sd.items()}, strict=True)
    return model


class ShiftRecorder:
    """Wraps random.randint inside the reference unet module to record the shift draws."""

    def __init__(self):
        self.draws = []
        self._orig = random.randint

    def __enter__(self):
        def rec(a, b):
            v = self._orig(a, b)
            self.draws.append(v)
            return v
        runet.random.randint = rec  # runet.random is the global `random` module
        return self

    def __exit__(self, *a):
        random.randint = self._orig


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if th.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ------------------------------------------------------------------ fixtures
def gen_tables():
    out = {}
    for sched in ("linear", "cosine"):
        for steps in (1000, 50):
            out[f"betas_{sched}_{steps}"] = rgd.get_named_beta_schedule(sched, steps)
    spacings = {}
    for steps, sc in ((1000, "2"), (1000, "250"), (1000, "ddim25"), (1000, "10,15,20"),
                      (300, "10,15,20"), (1000, "1000"), (50, "ddim10")):
        spacings[f"{steps}|{sc}"] = sorted(rresp.space_timesteps(steps, sc))
    with open(os.path.join(GOLD, "space_timesteps.json"), "w") as f:
        json.dump(spacings, f)
    for resp in ("", "2", "250", "ddim25"):
        diff = msu.create_gaussian_diffusion(steps=1000, timestep_respacing=resp)
        tag = resp or "full"
        for attr in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                     "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
            out[f"{tag}.{attr}"] = getattr(diff, attr)
        out[f"{tag}.timestep_map"] = np.asarray(diff.timestep_map)
    t = th.tensor([0, 1, 17, 999, 500], dtype=th.int64)
    out["temb_t"] = t
    out["temb_128"] = rnn.timestep_embedding(t, 128)
    out["temb_64_float"] = rnn.timestep_embedding(t.float() * 0.25, 64)
    save("tables", **out)


def gen_keys():
    res = {}
    for name in ("tiny", "full"):
        f = flags(name)
        model, _ = msu.create_model_and_diffusion(**f)
        res[name] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        res[name + "_nparams"] = sum(p.numel() for p in model.parameters())
    f = flags("tiny", learn_sigma=True)
    model, _ = msu.create_model_and_diffusion(**f)
    res["tiny_learn_sigma"] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as fo:
        json.dump(res, fo)
    print("wrote state_dict_keys.json", {k: (len(v) if isinstance(v, list) else v) for k, v in res.items()})


def _inputs(f, B, seed):
    g = th.Generator().manual_seed(seed)
    video = th.randn(B, *f["video_size"], generator=g)
    audio = th.randn(B, *f["audio_size"], generator=g)
    return video, audio


def gen_forward(name, B, seed, tsteps, **over):
    f = flags(name, **over)
    model, _ = msu.create_model_and_diffusion(**f)
    synth_init(model).eval()
    video, audio = _inputs(f, B, seed)
    t = th.tensor(tsteps, dtype=th.int64)
    random.seed(seed)
    with ShiftRecorder() as rec, th.no_grad():
        vo, ao = model(video, audio, t)
    tag = name + ("_ls" if over.get("learn_sigma") else "")
    save(f"{tag}_forward", seed=seed, B=B, t=t, shifts=np.asarray(rec.draws),
         video_out=vo, audio_out=ao)


def gen_blocks():
    """Per-module goldens straight from the reference classes (own small shapes)."""
    out = {}
    g = th.Generator().manual_seed(7)
    # --- ResBlock variants: (name, cin, cout, kwargs, F, H, L)
    specs = [
        ("res_plain", 64, 64, dict(audio_dilation=2), 8, 8, 256),
        ("res_widen", 64, 128, dict(audio_dilation=512), 8, 8, 256),   # dilation >= L
        ("res_down", 64, 64, dict(audio_dilation=8, down=True), 8, 8, 256),
        ("res_up", 64, 64, dict(audio_dilation=4, up=True), 8, 4, 64),
        ("res_attn", 128, 128, dict(audio_dilation=1, video_attention=True, audio_attention=True), 8, 4, 40),
    ]
    for nm, cin, cout, kw, F, H, L in specs:
        blk = runet.ResBlock(cin, 64, 0.0, out_channels=cout, use_scale_shift_norm=True,
                             num_heads=4, **kw)
        blk.load_state_dict({k: synth_tensor(nm + "." + k, v.shape) for k, v in blk.state_dict().items()})
        blk.eval()
        v = th.randn(2, F, cin, H, H, generator=g)
        a = th.randn(2, cin, L, generator=g)
        emb = th.randn(2, 64, generator=g)
        with th.no_grad():
            vo, ao = blk(v, a, emb)
        out[nm + ".video_in"], out[nm + ".audio_in"], out[nm + ".emb"] = v, a, emb
        out[nm + ".video_out"], out[nm + ".audio_out"] = vo, ao
    # --- plain (non scale-shift) ResBlock
    blk = runet.ResBlock(64, 64, 0.0, use_scale_shift_norm=False, audio_dilation=1)
    blk.load_state_dict({k: synth_tensor("res_noss." + k, v.shape) for k, v in blk.state_dict().items()})
    blk.eval()
    v = th.randn(1, 8, 64, 4, 4, generator=g); a = th.randn(1, 64, 32, generator=g); emb = th.randn(1, 64, generator=g)
    with th.no_grad():
        vo, ao = blk(v, a, emb)
    out.update({"res_noss.video_in": v, "res_noss.audio_in": a, "res_noss.emb": emb,
                "res_noss.video_out": vo, "res_noss.audio_out": ao})
    # --- CrossAttentionBlock: (name, C, head_ch, F, H, L, window, shift_flag, forced shift)
    xs = [
        ("xattn_w1_s0", 64, 32, 8, 4, 64, 1, True, 0),
        ("xattn_w1_s5", 64, 32, 8, 4, 64, 1, True, 5),
        ("xattn_w4_s3", 128, 32, 8, 4, 64, 4, True, 3),     # wraps: frames 3..6 for i=0, (7+3..)%8
        ("xattn_w8_s0", 64, 16, 8, 2, 8, 8, True, 0),       # 1 audio token per frame
        ("xattn_rem_s2", 64, 32, 16, 2, 100, 4, True, 2),   # L % F != 0: apf=6, remainder 4
        ("xattn_full", 64, 32, 8, 2, 32, 8, False, 0),      # middle-block style: window=F, no shift
    ]
    for nm, C, hc, F, H, L, win, sflag, shift in xs:
        blk = runet.CrossAttentionBlock(C, num_head_channels=hc, local_window=win, window_shift=sflag)
        blk.load_state_dict({k: synth_tensor(nm + "." + k, v.shape) for k, v in blk.state_dict().items()})
        blk.eval()
        v = th.randn(2, F, C, H, H, generator=g)
        a = th.randn(2, C, L, generator=g)
        orig = random.randint
        runet.random.randint = lambda lo, hi, s=shift: s
        try:
            with th.no_grad():
                vo, ao = blk(v, a)
        finally:
            random.randint = orig
        out[nm + ".video_in"], out[nm + ".audio_in"] = v, a
        out[nm + ".video_out"], out[nm + ".audio_out"] = vo, ao
        out[nm + ".meta"] = np.asarray([C, hc, F, H, L, win, int(sflag), shift])
    save("blocks", **out)


def gen_psample(name, B, seed, respacing, **over):
    f = flags(name, timestep_respacing=respacing, **over)
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).eval()
    shape = {"video": (B, *f["video_size"]), "audio": (B, *f["audio_size"])}
    th.manual_seed(seed)
    random.seed(seed)
    with ShiftRecorder() as rec:
        sample = diff.p_sample_loop(model, shape=shape, device=th.device("cpu"), progress=False,
                                    clip_denoised=True)
    tag = name + ("_ls" if over.get("learn_sigma") else "")
    save(f"{tag}_psample{respacing}", seed=seed, B=B, shifts=np.asarray(rec.draws),
         timestep_map=np.asarray(diff.timestep_map),
         video=sample["video"], audio=sample["audio"])


def gen_train_loss(name, B, seed, **over):
    f = flags(name, **{k: v for k, v in over.items() if not k.startswith("_")})
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).train()  # dropout p=0 -> deterministic
    g = th.Generator().manual_seed(seed)
    x0 = {"video": th.rand(B, *f["video_size"], generator=g) * 2 - 1,
          "audio": th.rand(B, *f["audio_size"], generator=g) * 2 - 1}
    noise = {"video": th.randn(B, *f["video_size"], generator=g),
             "audio": th.randn(B, *f["audio_size"], generator=g)}
    t = th.tensor([3, 977][:B], dtype=th.int64)
    random.seed(seed)
    with ShiftRecorder() as rec:
        losses = diff.multimodal_training_losses(model, x0, t, noise=noise)
    nfwd = len(rec.draws)
    # gradients need the backward recompute, which re-draws the shifts (nn.py:262-270): record them
    with ShiftRecorder() as rec2:
        losses["loss"].mean().backward()
    grads = {}
    for k in ("time_embed.0.weight", "input_blocks.0.0.video_conv.video_conv_spatial.weight",
              "middle_blocks.1.v_qkv.weight", "video_out.2.video_conv.bias", "audio_out.2.audio_conv.weight"):
        p = dict(model.named_parameters())[k]
        grads["grad." + k] = p.grad
    for k in over.get("_extra_grads", ()):
        grads["grad." + k] = dict(model.named_parameters())[k].grad
    tag = name + ("_ls" if over.get("learn_sigma") else "") + ("_nofilm" if over.get("use_scale_shift_norm") is False else "")
    save(f"{tag}_train_loss", seed=seed, B=B, t=t, shifts_fwd=np.asarray(rec.draws[:nfwd]),
         shifts_bwd=np.asarray(rec2.draws),
         **{k: v for k, v in losses.items()}, **grads)


def gen_train_grads_full(seed=71, stride=997):
    """configs[3] at the shipped size, batch 1: loss terms, the L2 norm of EVERY parameter's gradient, and a strided subsample
    (every `stride`-th element, flat order) of every gradient - the whole backward of multimodal_training_losses
    (multimodal_gaussian_diffusion.py:1114-1203) at full size in ~2 MB."""
    f = flags("full")
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).train()  # dropout p=0 -> deterministic
    g = th.Generator().manual_seed(seed)
    x0 = {"video": th.rand(1, *f["video_size"], generator=g) * 2 - 1, "audio": th.rand(1, *f["audio_size"], generator=g) * 2 - 1}
    noise = {"video": th.randn(1, *f["video_size"], generator=g), "audio": th.randn(1, *f["audio_size"], generator=g)}
    t = th.tensor([412], dtype=th.int64)
    random.seed(seed)
    with ShiftRecorder() as rec:
        losses = diff.multimodal_training_losses(model, x0, t, noise=noise)
    nfwd = len(rec.draws)
    with ShiftRecorder() as rec2:
        losses["loss"].mean().backward()
    names, norms, subs, offs = [], [], [], [0]
    for k, p in model.named_parameters():
        gr = p.grad.detach().flatten()
        names.append(k)
        norms.append(float(gr.double().norm()))
        subs.append(gr[::stride].clone())
        offs.append(offs[-1] + subs[-1].numel())
    save("full_train_grads", seed=seed, B=1, t=t, stride=stride, shifts_fwd=np.asarray(rec.draws[:nfwd]), shifts_bwd=np.asarray(rec2.draws),
         names=np.asarray(names), norms=np.asarray(norms), sub=th.cat(subs), sub_off=np.asarray(offs),
         **{k: v for k, v in losses.items()})


def gen_ddim(name, B, seed, respacing, eta):
    """ddim_sample_loop (gd:955-1046) final sample; the loop draws x_T and per-step noise even with eta = 0."""
    f = flags(name, timestep_respacing=respacing)
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).eval()
    shape = {"video": (B, *f["video_size"]), "audio": (B, *f["audio_size"])}
    th.manual_seed(seed)
    random.seed(seed)
    with ShiftRecorder() as rec:
        sample = diff.ddim_sample_loop(model, shape=shape, device=th.device("cpu"), progress=False, clip_denoised=True, eta=eta)
    save(f"{name}_ddim{respacing}_eta{int(eta * 10):02d}", seed=seed, B=B, eta=eta, shifts=np.asarray(rec.draws),
         video=sample["video"], audio=sample["audio"])


def gen_cond(name, B, seed, respacing, which, class_scale):
    """conditional_p_sample_loop (gd:584-819): replacement (class_scale 0) or gradient-guided sampling given one stream."""
    f = flags(name, timestep_respacing=respacing)
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).eval()
    shape = {"video": (B, *f["video_size"]), "audio": (B, *f["audio_size"])}
    g = th.Generator().manual_seed(seed + 1000)
    cond = th.rand(*shape[which], generator=g) * 2 - 1
    th.manual_seed(seed)
    random.seed(seed)
    with ShiftRecorder() as rec:
        sample = diff.conditional_p_sample_loop(model, shape=shape, use_fp16=False, model_kwargs={which: cond.clone()},
                                                device=th.device("cpu"), progress=False, clip_denoised=True, class_scale=class_scale)
    save(f"{name}_cond_{which}_{'guided' if class_scale else 'replace'}{respacing}", seed=seed, B=B, class_scale=class_scale,
         shifts=np.asarray(rec.draws), cond=cond, video=sample["video"].detach(), audio=sample["audio"].detach())


def gen_dpm(tag, seed, predict_x0, thresholding, config="tiny", **sample_kw):
    """Multimodal DPM-Solver(++) sample() on the tiny config (or, round 6, the shipped base model: BASELINE configs[4] at full size),
    B = 2 (B = 1 fails inside the reference, dpm:344-345)."""
    from ref_mm import multimodal_dpm_solver_plus as rdpm
    f = flags(config)
    model, diff = msu.create_model_and_diffusion(**f)
    synth_init(model).eval()
    B = 2
    th.manual_seed(seed)
    random.seed(seed)
    x_T = {"video": th.randn(B, *f["video_size"]), "audio": th.randn(B, *f["audio_size"])}
    solver = rdpm.DPM_Solver(model=model, alphas_cumprod=th.tensor(diff.alphas_cumprod, dtype=th.float32), predict_x0=predict_x0,
                             thresholding=thresholding)
    import contextlib, io
    with ShiftRecorder() as rec, contextlib.redirect_stdout(io.StringIO()):
        out = solver.sample({k: v.clone() for k, v in x_T.items()}, **sample_kw)
    per_fwd = 9 if config == "tiny" else 15          # shift draws per forward: one per shifted CrossAttentionBlock
    save(tag, seed=seed, B=B, shifts=np.asarray(rec.draws), nfe=len(rec.draws) // per_fwd, video=out["video"], audio=out["audio"])


def gen_noise_schedule():
    """The reference's NoiseScheduleVP (dpm:11-181) on fixed times: the three schedule families, both parameterisations of the
    discrete one; marginal_* and inverse_lambda (the solver's integer model timesteps are floor((t - 1/N) N): t must match to the bit)."""
    from ref_mm import multimodal_dpm_solver_plus as rdpm
    betas = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
    g = th.Generator().manual_seed(5)
    out = {}
    for tag, kw in (("discrete_betas", dict(schedule="discrete", betas=th.tensor(betas))),
                    ("discrete_acp", dict(schedule="discrete", alphas_cumprod=th.tensor(np.cumprod(1 - betas), dtype=th.float32))),
                    ("linear", dict(schedule="linear")), ("cosine", dict(schedule="cosine"))):
        ns = rdpm.NoiseScheduleVP(**kw)
        t = th.cat([th.rand(61, generator=g) * (ns.T - 2e-3) + 1e-3, th.tensor([1e-3, 1. / 1000, 0.5, ns.T])]).float()
        lam = ns.marginal_lambda(t)
        out.update({f"{tag}_t": t, f"{tag}_log_alpha": ns.marginal_log_mean_coeff(t), f"{tag}_alpha": ns.marginal_alpha(t),
                    f"{tag}_std": ns.marginal_std(t), f"{tag}_lambda": lam, f"{tag}_inv": ns.inverse_lambda(lam),
                    f"{tag}_T": ns.T, f"{tag}_N": ns.total_N})
    save("noise_schedule_vp", **out)


SR_TINY = dict(large_size=64, small_size=16, sr_num_channels=32, sr_num_res_blocks=1, sr_attention_resolutions="2,4", sr_num_heads=2,
               sr_resblock_updown=True)


def _sr_build(**over):
    from ref_mm import script_util as rsu
    d = rsu.image_sr_model_and_diffusion_defaults()
    d.update(SR_TINY)
    d.update(over)
    model, diff = rsu.image_sr_create_model_and_diffusion(**d)
    synth_init(model).eval()
    return d, model, diff


def gen_sr():
    """Image super-resolution U-Net (image_unet.ImageSuperResModel) + its tensor-valued diffusion: state-dict keys, one forward,
    a DDIM loop and a DDPM loop with the clip-repeated start noise of multimodal_sample_sr.py:191-196."""
    d, model, diff = _sr_build()
    with open(os.path.join(GOLD, "sr_state_dict_keys.json"), "w") as f:
        json.dump({"tiny": [[k, list(v.shape)] for k, v in model.state_dict().items()]}, f)
    g = th.Generator().manual_seed(71)
    B = 2
    x = th.randn(B, 3, 64, 64, generator=g)
    low = th.rand(B, 3, 16, 16, generator=g) * 2 - 1
    t = th.tensor([3, 977])
    with th.no_grad():
        y = model(x, t, low_res=low)
    save("sr_tiny_forward", x=x, low=low, t=t, y=y)
    for tag, resp, fn, kw in (("sr_tiny_ddim4", "ddim4", "ddim_sample_loop", {}), ("sr_tiny_ddpm3", "3", "p_sample_loop", dict(progress=False))):
        d, model, diff = _sr_build(sr_timestep_respacing=resp)
        th.manual_seed(72)
        noise = th.randn(1, 3, 64, 64).repeat(B, 1, 1, 1)
        with th.no_grad():
            out = getattr(diff, fn)(model, (B, 3, 64, 64), clip_denoised=True, model_kwargs={"low_res": low}, noise=noise.clone(),
                                    device=th.device("cpu"), **kw)
        save(tag, low=low, noise=noise, sample=out, timestep_map=np.asarray(diff.timestep_map))


def gen_sr_dpm():
    """Single-modal DPM-Solver(++) on the tiny SR model (multimodal_sample_sr.py:199-228: multistep order 2, time_uniform)."""
    from ref_mm import dpm_solver_plus as rsdpm
    d, model, diff = _sr_build()
    g = th.Generator().manual_seed(73)
    B = 2
    low = th.rand(B, 3, 16, 16, generator=g) * 2 - 1
    noise = th.randn(1, 3, 64, 64, generator=g).repeat(B, 1, 1, 1)
    for tag, px0 in (("sr_tiny_dpm_multistep2", False), ("sr_tiny_dpmpp_multistep2", True)):
        solver = rsdpm.DPM_Solver(model=model, alphas_cumprod=th.tensor(diff.alphas_cumprod, dtype=th.float32), predict_x0=px0,
                                  model_kwargs={"low_res": low})
        out = solver.sample(noise.clone(), steps=6, order=2, skip_type="time_uniform", method="multistep")
        save(tag, low=low, noise=noise, sample=out)


def gen_helpers():
    """q_mean_variance / q_posterior_mean_variance / _predict_* (gd:170-229,345-366) on random inputs."""
    f = flags("tiny", timestep_respacing="")
    _, diff = msu.create_model_and_diffusion(**f)
    g = th.Generator().manual_seed(77)
    a, b = th.randn(3, 4, 5, generator=g), th.randn(3, 4, 5, generator=g)
    t = th.tensor([0, 500, 999])
    qm, qv, qlv = diff.q_mean_variance(a, t)
    pm, pv, plv = diff.q_posterior_mean_variance(a, b, t)
    save("helpers", a=a, b=b, t=t, q_mean=qm, q_var=qv, q_logvar=qlv, post_mean=pm, post_var=pv, post_logvar=plv,
         xstart_from_eps=diff._predict_xstart_from_eps(a, t, b), xstart_from_xprev=diff._predict_xstart_from_xprev(a, t, b),
         eps_from_xstart=diff._predict_eps_from_xstart(a, t, b))


def gen_pmv_denoised():
    """p_mean_variance / p_sample of the reference (gd:231-343, 415-474) with a denoised_fn (x -> 0.5 x + 0.1, applied before the clamp) on a
    stand-in model that returns fixed random outputs: learned-range variance, respacing 10, clip on and off."""
    f = flags("tiny", timestep_respacing="10", learn_sigma=True)
    _, diff = msu.create_model_and_diffusion(**f)
    g = th.Generator().manual_seed(91)
    B = 3
    x = {"video": th.randn(B, 4, 3, 6, 6, generator=g), "audio": th.randn(B, 1, 40, generator=g)}
    vo, ao = th.randn(B, 4, 6, 6, 6, generator=g), th.randn(B, 2, 40, generator=g)
    t = th.tensor([9, 0, 4])
    model = lambda v, a, ts, **kw: (vo, ao)      # noqa: E731
    fn = lambda z: 0.5 * z + 0.1                  # noqa: E731
    arrs = dict(xv=x["video"], xa=x["audio"], vo=vo, ao=ao, t=t)
    for clip in (True, False):
        out = diff.p_mean_variance(model, x, t, clip_denoised=clip, denoised_fn=fn)
        for k in ("mean", "log_variance", "pred_xstart"):
            for key in ("video", "audio"):
                arrs[f"{k}_{key}_clip{int(clip)}"] = out[k][key]
    th.manual_seed(92)
    ps = diff.p_sample(model, x, t, clip_denoised=True, denoised_fn=fn)
    th.manual_seed(92)
    arrs["noise_v"], arrs["noise_a"] = th.randn_like(x["video"]), th.randn_like(x["audio"])
    arrs["sample_video"], arrs["sample_audio"] = ps["sample"]["video"], ps["sample"]["audio"]
    save("pmv_denoised", **arrs)


ALL = {
    "pmv_denoised": gen_pmv_denoised,
    "tables": gen_tables,
    "keys": gen_keys,
    "blocks": gen_blocks,
    "tiny_forward": lambda: gen_forward("tiny", 2, 11, [7, 812]),
    "tiny_ls_forward": lambda: gen_forward("tiny", 2, 12, [0, 999], learn_sigma=True),
    "mid_forward": lambda: gen_forward("mid", 1, 13, [431]),
    # the shipped base model at full size, one sample: pins the kernels that only run at the full model's shapes (fused VideoConv,
    # fused temporal attention, GEMM + gather head, ...) against the reference at the shapes bench.py times (round 5)
    "full_forward": lambda: gen_forward("full", 1, 14, [417]),
    "tiny_psample": lambda: gen_psample("tiny", 2, 21, "2"),
    "tiny_psample4": lambda: gen_psample("tiny", 1, 22, "4"),
    "tiny_ls_psample": lambda: gen_psample("tiny", 2, 23, "2", learn_sigma=True),
    "full_psample": lambda: gen_psample("full", 1, 0, "2"),
    "tiny_ddim_eta0": lambda: gen_ddim("tiny", 2, 41, "4", 0.0),
    "tiny_ddim_eta5": lambda: gen_ddim("tiny", 1, 42, "4", 0.5),
    "tiny_cond_replace": lambda: gen_cond("tiny", 2, 51, "4", "video", 0.0),
    "tiny_cond_guided_v": lambda: gen_cond("tiny", 1, 52, "4", "video", 3.0),
    "tiny_cond_guided_a": lambda: gen_cond("tiny", 1, 53, "2", "audio", 3.0),
    "helpers": gen_helpers,
    "noise_schedule": gen_noise_schedule,
    "sr": gen_sr,
    "sr_dpm": gen_sr_dpm,
    "dpm_singlestep3": lambda: gen_dpm("tiny_dpm_singlestep3", 61, False, False, steps=20, order=3, skip_type="logSNR", method="singlestep"),
    "dpm_singlestep2": lambda: gen_dpm("tiny_dpm_singlestep2", 62, False, False, steps=7, order=2, skip_type="time_quadratic", method="singlestep"),
    "dpm_multistep2": lambda: gen_dpm("tiny_dpm_multistep2", 63, False, False, steps=10, order=2, skip_type="time_uniform", method="multistep"),
    "dpmpp_multistep2": lambda: gen_dpm("tiny_dpmpp_multistep2", 64, True, True, steps=10, order=2, skip_type="logSNR", method="multistep",
                                        denoise=True),
    "dpmpp_adaptive2": lambda: gen_dpm("tiny_dpmpp_adaptive2", 65, True, True, steps=20, order=2, skip_type="logSNR", method="adaptive"),
    "dpm_adaptive3": lambda: gen_dpm("tiny_dpm_adaptive3", 66, False, False, order=3, method="adaptive", atol=0.05, rtol=0.1),
    # BASELINE configs[4], base-model half, at FULL size against the reference itself (round 6): DPM-Solver++ (predict_x0 + dynamic
    # thresholding) multistep order 2, 10 network evaluations, batch 2
    "full_dpmpp_multistep2": lambda: gen_dpm("full_dpmpp_multistep2", 67, True, True, config="full", steps=10, order=2, skip_type="logSNR",
                                             method="multistep"),
    "full_train_grads": gen_train_grads_full,
    "tiny_train_loss": lambda: gen_train_loss("tiny", 2, 31),
    "tiny_ls_train_loss": lambda: gen_train_loss("tiny", 2, 32, learn_sigma=True),
    # non-FiLM ResBlocks (use_scale_shift_norm=False, unet:473-477): h + emb_out, then the plain norm
    "tiny_nofilm_train_loss": lambda: gen_train_loss("tiny", 2, 33, use_scale_shift_norm=False,
                                                     _extra_grads=("input_blocks.1.0.emb_layers.1.weight", "middle_blocks.0.emb_layers.1.bias")),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    for k, fn in ALL.items():
        if a.only and k not in a.only:
            continue
        print("==", k)
        fn()
