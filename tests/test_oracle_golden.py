"""Pin the oracle (oracle/*.py) against fixtures captured from the imported reference.

CPU only.  Tolerances: the oracle and the reference are both fp32 torch on CPU but use
different op decompositions (conv3d vs conv2d+conv1d, arithmetic windows vs gathers), so
agreement is to fp32 round-off: rel-L2 <= 2e-5 per forward, <= 1e-4 after sampling loops.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

from helpers import GOLD, flags, gold, gold_keys, inputs, rel_l2, synth_sd
from oracle import diffusion_ref as dref
from oracle import unet_ref as uref


def test_arch_matches_reference_keys():
    """Every state-dict key the reference creates is consumed by the oracle's arch table."""
    keys = gold_keys()
    for name, fl in (("tiny", flags("tiny")), ("full", flags("full"))):
        cfg = uref.parse_cfg(fl)
        ins, mid, outs = uref.build_arch(cfg)
        prefixes = [l["prefix"] for blk in ins + [mid] + outs for l in blk]
        refkeys = [k for k, _ in keys[name]]
        top = {"time_embed", "video_out", "audio_out"}
        for k in refkeys:
            if k.split(".")[0] in top:
                continue
            assert any(k.startswith(p + ".") for p in prefixes), k
        for p in prefixes:
            assert any(k.startswith(p + ".") for k in refkeys), p
    assert keys["full_nparams"] == 133675524 and len(keys["full"]) == 1046


def test_tables_and_spacing():
    g = gold("tables")
    for sched in ("linear", "cosine"):
        for steps in (1000, 50):
            np.testing.assert_allclose(dref.beta_schedule(sched, steps), g[f"betas_{sched}_{steps}"], rtol=0, atol=0)
    with open(os.path.join(GOLD, "space_timesteps.json")) as f:
        sp = json.load(f)
    for k, v in sp.items():
        steps, sc = k.split("|")
        assert dref.space_timesteps(int(steps), sc) == v, k
    for resp in ("", "2", "250", "ddim25"):
        S = dref.Schedule(respacing=resp)
        tag = resp or "full"
        assert S.timestep_map == list(g[f"{tag}.timestep_map"])
        for mine, theirs in (("betas", "betas"), ("alphas_cumprod", "alphas_cumprod"),
                             ("sqrt_recip_ac", "sqrt_recip_alphas_cumprod"),
                             ("sqrt_recipm1_ac", "sqrt_recipm1_alphas_cumprod"),
                             ("post_var", "posterior_variance"),
                             ("post_logvar_clipped", "posterior_log_variance_clipped"),
                             ("post_c1", "posterior_mean_coef1"), ("post_c2", "posterior_mean_coef2"),
                             ("sqrt_ac", "sqrt_alphas_cumprod"), ("sqrt_1mac", "sqrt_one_minus_alphas_cumprod")):
            np.testing.assert_array_equal(getattr(S, mine), g[f"{tag}.{theirs}"])
    with pytest.raises(ValueError):
        dref.space_timesteps(10, "20")
    t = torch.from_numpy(g["temb_t"])
    assert rel_l2(uref.timestep_embedding(t, 128), g["temb_128"]) < 1e-6
    assert rel_l2(uref.timestep_embedding(t.float() * 0.25, 64), g["temb_64_float"]) < 1e-6


def _block_sd(g, nm, blk_keys):
    from mm_diffusion.synth import synth_tensor
    return {("B." + k): synth_tensor(nm + "." + k, s) for k, s in blk_keys}


RES_KEYS = None


def _res_keys(cin, cout, vattn, aattn, ss=True):
    ks = []
    def gn(p, c): ks.extend([(p + ".GroupNorm.weight", (c,)), (p + ".GroupNorm.bias", (c,))])
    def cv(p, shape): ks.extend([(p + ".weight", shape), (p + ".bias", (shape[0],))])
    gn("video_in_layers.0", cin); cv("video_in_layers.2.video_conv_spatial", (cout, cin, 3, 3)); cv("video_in_layers.2.video_conv_temporal", (cout, cout, 3))
    gn("audio_in_layers.0", cin); cv("audio_in_layers.2.audio_conv", (cout, cin, 3))
    cv("emb_layers.1", (2 * cout if ss else cout, 64))
    gn("video_out_layers.0", cout); cv("video_out_layers.3.video_conv", (cout, cout, 1, 1, 1))
    gn("audio_out_layers.0", cout); cv("audio_out_layers.3.audio_conv", (cout, cout, 1))
    if cin != cout:
        cv("video_skip_connection.video_conv", (cout, cin, 1, 1, 1)); cv("audio_skip_connection.audio_conv", (cout, cin, 1))
    for nm, on in (("spatial_attention_block", vattn), ("temporal_attention_block", vattn), ("audio_attention_block", aattn)):
        if on:
            gn(nm + ".norm", cout); cv(nm + ".qkv", (3 * cout, cout, 1)); cv(nm + ".proj_out", (cout, cout, 1))
    return ks


@pytest.mark.parametrize("nm,cin,cout,kw", [
    ("res_plain", 64, 64, dict(dilation=2)),
    ("res_widen", 64, 128, dict(dilation=512)),
    ("res_down", 64, 64, dict(dilation=8, down=True)),
    ("res_up", 64, 64, dict(dilation=4, up=True)),
    ("res_attn", 128, 128, dict(dilation=1, vattn=True, aattn=True)),
    ("res_noss", 64, 64, dict(dilation=1, ss=False)),
])
def test_resblock(nm, cin, cout, kw):
    g = gold("blocks")
    ss = kw.pop("ss", True)
    layer = dict(kind="res", prefix="B", cin=cin, cout=cout, up=False, down=False, vattn=False, aattn=False)
    layer.update(kw)
    sd = _block_sd(g, nm, _res_keys(cin, cout, layer["vattn"], layer["aattn"], ss))
    cfg = dict(use_scale_shift_norm=ss, num_heads=4)
    v = torch.from_numpy(g[nm + ".video_in"]).permute(0, 2, 1, 3, 4)
    a = torch.from_numpy(g[nm + ".audio_in"])
    emb = torch.from_numpy(g[nm + ".emb"])
    with torch.no_grad():
        vo, ao = uref.res_block(v, a, emb, sd, layer, cfg)
    assert rel_l2(vo.permute(0, 2, 1, 3, 4), g[nm + ".video_out"]) < 1e-5
    assert rel_l2(ao, g[nm + ".audio_out"]) < 1e-5


@pytest.mark.parametrize("nm", ["xattn_w1_s0", "xattn_w1_s5", "xattn_w4_s3", "xattn_w8_s0", "xattn_rem_s2", "xattn_full"])
def test_cross_attention(nm):
    g = gold("blocks")
    C, hc, F, H, L, win, sflag, shift = [int(x) for x in g[nm + ".meta"]]
    ks = []
    for p in ("v_norm", "a_norm"):
        ks += [(p + ".GroupNorm.weight", (C,)), (p + ".GroupNorm.bias", (C,))]
    for p in ("v_qkv", "a_qkv"):
        ks += [(p + ".weight", (3 * C, C, 1)), (p + ".bias", (3 * C,))]
    ks += [("video_proj_out.video_conv.weight", (C, C, 1, 1, 1)), ("video_proj_out.video_conv.bias", (C,)),
           ("audio_proj_out.audio_conv.weight", (C, C, 1)), ("audio_proj_out.audio_conv.bias", (C,))]
    sd = _block_sd(g, nm, ks)
    layer = dict(kind="cross", prefix="B", ch=C, heads=C // hc, window=win, shift=bool(sflag))
    v = torch.from_numpy(g[nm + ".video_in"]).permute(0, 2, 1, 3, 4)
    a = torch.from_numpy(g[nm + ".audio_in"])
    with torch.no_grad():
        vo, ao = uref.cross_attention(v, a, sd, layer, shift)
    assert rel_l2(vo.permute(0, 2, 1, 3, 4), g[nm + ".video_out"]) < 1e-5
    assert rel_l2(ao, g[nm + ".audio_out"]) < 1e-5


@pytest.mark.parametrize("tag,cfgname,keyset,over", [
    ("tiny", "tiny", "tiny", {}),
    ("tiny_ls", "tiny", "tiny_learn_sigma", dict(learn_sigma=True)),
    ("mid", "mid", "tiny", {}),          # mid shares the tiny parameter shapes (same channels)
    ("full", "full", "full", {}),        # the shipped base model at full size, one sample (round 5)
])
def test_forward(tag, cfgname, keyset, over):
    g = gold(tag + "_forward")
    fl = flags(cfgname, **over)
    sd = synth_sd(keyset)
    video, audio = inputs(fl, int(g["B"]), int(g["seed"]))
    # (1) replaying the recorded shifts, (2) drawing them from random.seed like the reference
    vo, ao = uref.unet_forward(sd, uref.parse_cfg(fl), video, audio, torch.from_numpy(g["t"]), shifts=list(g["shifts"]))
    assert rel_l2(vo, g["video_out"]) < 2e-5 and rel_l2(ao, g["audio_out"]) < 2e-5
    random.seed(int(g["seed"]))
    vo2, ao2 = uref.unet_forward(sd, uref.parse_cfg(fl), video, audio, torch.from_numpy(g["t"]))
    assert torch.equal(vo, vo2) and torch.equal(ao, ao2)


@pytest.mark.parametrize("tag,cfgname,keyset,resp,over", [
    ("tiny_psample2", "tiny", "tiny", "2", {}),
    ("tiny_psample4", "tiny", "tiny", "4", {}),
    ("tiny_ls_psample2", "tiny", "tiny_learn_sigma", "2", dict(learn_sigma=True)),
])
def test_psample_loop(tag, cfgname, keyset, resp, over):
    g = gold(tag)
    fl = flags(cfgname, **over)
    B = int(g["B"])
    S = dref.Schedule(respacing=resp, learn_sigma=bool(over.get("learn_sigma")))
    assert S.timestep_map == list(g["timestep_map"])
    model = uref.OracleModel(synth_sd(keyset), fl, shifts=list(g["shifts"]))
    torch.manual_seed(int(g["seed"]))
    x = dref.p_sample_loop(S, model, {"video": (B, *fl["video_size"]), "audio": (B, *fl["audio_size"])})
    assert rel_l2(x["video"], g["video"]) < 1e-4 and rel_l2(x["audio"], g["audio"]) < 1e-4


@pytest.mark.parametrize("tag,resp,eta", [("tiny_ddim4_eta00", "4", 0.0), ("tiny_ddim4_eta05", "4", 0.5)])
def test_ddim_loop(tag, resp, eta):
    g = gold(tag)
    fl = flags("tiny")
    B = int(g["B"])
    S = dref.Schedule(respacing=resp)
    model = uref.OracleModel(synth_sd("tiny"), fl, shifts=list(g["shifts"]))
    torch.manual_seed(int(g["seed"]))
    x = dref.ddim_sample_loop(S, model, {"video": (B, *fl["video_size"]), "audio": (B, *fl["audio_size"])}, eta=eta)
    assert rel_l2(x["video"], g["video"]) < 2e-4 and rel_l2(x["audio"], g["audio"]) < 2e-4


def test_conditional_replacement_loop():
    g = gold("tiny_cond_video_replace4")
    fl = flags("tiny")
    B = int(g["B"])
    S = dref.Schedule(respacing="4")
    model = uref.OracleModel(synth_sd("tiny"), fl, shifts=list(g["shifts"]))
    torch.manual_seed(int(g["seed"]))
    x = dref.cond_replace_loop(S, model, {"video": (B, *fl["video_size"]), "audio": (B, *fl["audio_size"])},
                               {"video": torch.from_numpy(g["cond"])})
    assert rel_l2(x["video"], g["video"]) < 2e-4 and rel_l2(x["audio"], g["audio"]) < 2e-4


def test_posterior_and_predict_helpers():
    g = gold("helpers")
    S = dref.Schedule()
    a, b, t = torch.from_numpy(g["a"]), torch.from_numpy(g["b"]), torch.from_numpy(g["t"])
    pm, pv, plv = dref.q_posterior(S, a, b, t)
    for got, key in ((pm, "post_mean"), (pv, "post_var"), (plv, "post_logvar"),
                     (dref.predict_xstart_from_eps(S, a, t, b), "xstart_from_eps"),
                     (dref.predict_xstart_from_xprev(S, a, t, b), "xstart_from_xprev"),
                     (dref.predict_eps_from_xstart(S, a, t, b), "eps_from_xstart")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=1e-6)


DPM_CASES = {
    "tiny_dpm_singlestep3": (False, dict(steps=20, order=3, skip_type="logSNR", method="singlestep")),
    "tiny_dpm_singlestep2": (False, dict(steps=7, order=2, skip_type="time_quadratic", method="singlestep")),
    "tiny_dpm_multistep2": (False, dict(steps=10, order=2, skip_type="time_uniform", method="multistep")),
    "tiny_dpmpp_multistep2": (True, dict(steps=10, order=2, skip_type="logSNR", method="multistep", denoise=True)),
    "tiny_dpmpp_adaptive2": (True, dict(order=2, method="adaptive")),
}


@pytest.mark.parametrize("tag", list(DPM_CASES))
def test_dpm_solver(tag):
    """Oracle restatement of the multimodal DPM-Solver(++) driver vs the reference's own sample() output."""
    from oracle import dpm_ref
    g = gold(tag)
    pp, kw = DPM_CASES[tag]
    fl = flags("tiny")
    B = int(g["B"])
    model = uref.OracleModel(synth_sd("tiny"), fl, shifts=list(g["shifts"]))
    torch.manual_seed(int(g["seed"]))
    x_T = {"video": torch.randn(B, *fl["video_size"]), "audio": torch.randn(B, *fl["audio_size"])}
    S = dref.Schedule()
    solver = dpm_ref.Solver(model, torch.tensor(S.alphas_cumprod, dtype=torch.float32), predict_x0=pp, thresholding=pp)
    out = solver.sample(x_T, **kw)
    assert rel_l2(out["video"], g["video"]) < 1e-3 and rel_l2(out["audio"], g["audio"]) < 1e-3


SR_CFG = dict(model_channels=32, channel_mult=(1, 2, 3, 4), num_res_blocks=1, attention_ds=(2, 4), heads=2)


def _sr_sd():
    with open(os.path.join(GOLD, "sr_state_dict_keys.json")) as f:
        keys = json.load(f)["tiny"]
    from mm_diffusion.synth import synth_tensor
    return {k: synth_tensor(k, shp) for k, shp in keys}


def test_sr_unet_forward_and_loops():
    """Oracle restatement of the image SR U-Net and its tensor-valued DDIM / DDPM loops vs the reference's outputs."""
    from oracle import sr_ref
    sd = _sr_sd()
    g = gold("sr_tiny_forward")
    y = sr_ref.sr_forward(sd, SR_CFG, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["low"]))
    assert rel_l2(y, g["y"]) < 1e-4
    for tag, resp, fn in (("sr_tiny_ddim4", "ddim4", sr_ref.ddim_sample_loop), ("sr_tiny_ddpm3", "3", sr_ref.p_sample_loop)):
        g = gold(tag)
        S = dref.Schedule(respacing=resp, learn_sigma=True)
        assert S.timestep_map == list(g["timestep_map"])
        low, noise = torch.from_numpy(g["low"]), torch.from_numpy(g["noise"])
        model = lambda x, t: sr_ref.sr_forward(sd, SR_CFG, x, t, low)      # noqa: E731
        torch.manual_seed(72)
        torch.randn(1, 3, 64, 64)                                          # the fixture drew its start noise from this seed first
        out = fn(S, model, noise.shape, noise)
        assert rel_l2(out, g["sample"]) < 2e-4


@pytest.mark.parametrize("tag,px0", [("sr_tiny_dpm_multistep2", False), ("sr_tiny_dpmpp_multistep2", True)])
def test_sr_single_modal_dpm_solver(tag, px0):
    from oracle import dpm_ref, sr_ref
    g = gold(tag)
    sd = _sr_sd()
    low, noise = torch.from_numpy(g["low"]), torch.from_numpy(g["noise"])
    model = lambda x, t: sr_ref.sr_forward(sd, SR_CFG, x, t, low)      # noqa: E731
    solver = dpm_ref.Solver(model, torch.tensor(dref.Schedule().alphas_cumprod, dtype=torch.float32), predict_x0=px0, single=True)
    out = solver.sample({"x": noise}, steps=6, order=2, skip_type="time_uniform", method="multistep")["x"]
    assert rel_l2(out, g["sample"]) < 1e-3


def test_full_config1_two_step():
    """BASELINE config[0]: Landscape base model, batch 1, 2-step DDPM on the CPU path."""
    g = gold("full_psample2")
    fl = flags("full")
    S = dref.Schedule(respacing="2")
    assert S.timestep_map == [0, 999]
    model = uref.OracleModel(synth_sd("full"), fl, shifts=list(g["shifts"]))
    torch.manual_seed(int(g["seed"]))
    x = dref.p_sample_loop(S, model, {"video": (1, *fl["video_size"]), "audio": (1, *fl["audio_size"])})
    assert x["video"].shape == (1, 16, 3, 64, 64) and x["audio"].shape == (1, 1, 25600)
    assert rel_l2(x["video"], g["video"]) < 2e-4 and rel_l2(x["audio"], g["audio"]) < 2e-4


@pytest.mark.parametrize("tag,keyset,over", [("tiny", "tiny", {}), ("tiny_ls", "tiny_learn_sigma", dict(learn_sigma=True))])
def test_training_losses(tag, keyset, over):
    g = gold(tag + "_train_loss")
    fl = flags("tiny", **over)
    B, seed = int(g["B"]), int(g["seed"])
    gen = torch.Generator().manual_seed(seed)
    x0 = {"video": torch.rand(B, *fl["video_size"], generator=gen) * 2 - 1,
          "audio": torch.rand(B, *fl["audio_size"], generator=gen) * 2 - 1}
    noise = {"video": torch.randn(B, *fl["video_size"], generator=gen),
             "audio": torch.randn(B, *fl["audio_size"], generator=gen)}
    S = dref.Schedule(learn_sigma=bool(over.get("learn_sigma")))
    model = uref.OracleModel(synth_sd(keyset), fl, shifts=list(g["shifts_fwd"]))
    terms = dref.training_losses(S, model, x0, torch.from_numpy(g["t"]), noise)
    for k in ("loss", "mse_video", "mse_audio") + (("vb_video", "vb_audio") if over else ()):
        np.testing.assert_allclose(terms[k].numpy(), g[k], rtol=2e-4, atol=1e-6)


def test_p_mean_variance_with_denoised_fn():
    """oracle p_mean_variance(denoised_fn=...) against the reference's p_mean_variance / p_sample run with the same function on a stand-in
    model (fixture pmv_denoised.npz, tools/gen_golden.py: gen_pmv_denoised): the function acts on the x_0 prediction BEFORE the clamp."""
    g = gold("pmv_denoised")
    S = dref.Schedule(respacing="10", learn_sigma=True)
    t = torch.from_numpy(g["t"])
    fn = lambda z: 0.5 * z + 0.1      # noqa: E731
    for key, xk, ok, cdim in (("video", "xv", "vo", 2), ("audio", "xa", "ao", 1)):
        x, o = torch.from_numpy(g[xk]), torch.from_numpy(g[ok])
        for clip in (1, 0):
            mean, logvar, x0 = dref.p_mean_variance(S, o, x, t, cdim, clip=bool(clip), denoised_fn=fn)
            assert rel_l2(mean, g[f"mean_{key}_clip{clip}"]) < 1e-6
            assert rel_l2(logvar, g[f"log_variance_{key}_clip{clip}"]) < 1e-6
            assert rel_l2(x0, g[f"pred_xstart_{key}_clip{clip}"]) < 1e-6
        mean, logvar, _ = dref.p_mean_variance(S, o, x, t, cdim, clip=True, denoised_fn=fn)
        nz = (t != 0).float().reshape(-1, *([1] * (x.dim() - 1)))
        noise = torch.from_numpy(g["noise_v" if key == "video" else "noise_a"])
        assert rel_l2(mean + nz * torch.exp(0.5 * logvar) * noise, g[f"sample_{key}"]) < 1e-6
