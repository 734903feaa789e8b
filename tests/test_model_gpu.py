"""End-to-end parity of the HIP path (through libmmd) against golden fixtures captured from the reference and
against the oracle, on the same seeded inputs / shifts / noise.

Tolerances (rel-L2 vs the fp32 CPU reference):
  fp32 mode : forward <= 1e-4, sampling loops <= 5e-4      (SURVEY 8c suggested bounds; exact-fp32 MFMA)
  bf16 mode : forward <= 3e-2, sampling loops <= 1e-1       (no low-precision oracle exists upstream, H6; the
              eps -> x0 map multiplies the model error by sqrt(1/ac_t - 1) ~ 157 at t=999 before the clamp,
              so a loop sees the ~1.6e-2 forward error amplified; measured 5-7e-2 on the 2-step fixtures)
"""
import numpy as np
import pytest
import torch

from helpers import flags, gold, inputs, rel_l2, synth_sd

pytestmark = pytest.mark.gpu

# Stated bounds = ~1.5x the distances measured on the MI355X at the round-3 build (profiles/r03_parity_model_tests.txt; the outputs are
# bitwise repeatable, so the margin is for future kernel changes, not for noise).  Forward: fp32 1.8e-6 ... 2.4e-6, bf16 1.3e-2 ... 1.6e-2.
FWD_TOL = {torch.float32: 2e-5, torch.bfloat16: 2.5e-2}
# Loops, per fixture (video / audio measured): the epsilon -> x0 map multiplies the model error by sqrt(1 / abar_t - 1) ~ 157 at t = 999, so
# the 2-step loops (t = 999, 499) sit far above the 4-step one; learn_sigma adds the interpolated log-variance of a bf16 output.
LOOP_TOL = {torch.float32: 1e-4, torch.bfloat16: 7.5e-2}      # full-size 2-step: fp32 2.4e-5 / 1.5e-5, bf16 4.9e-2 / 4.8e-2
LOOP_TOL_BF16 = {"tiny_psample2": 9e-2,                      # 5.4e-2 / 6.4e-2
                 "tiny_psample4": 3e-2,                      # 1.7e-2 / 1.6e-2
                 "tiny_ls_psample2": 1.2e-1}                 # 7.9e-2 / 8.5e-2   (fp32, all three: 3e-6 ... 1.7e-5)


def build(cfgname, keyset, dt, **over):
    from mm_diffusion import logger, multimodal_script_util as msu
    logger.set_quiet(True)
    fl = flags(cfgname, use_fp16=(dt == torch.bfloat16), **over)
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd(keyset))
    model.cuda().eval()
    assert model.dtype == dt
    return fl, model, diff


def replay(model, shifts):
    it = iter(int(s) for s in shifts)
    model.shift_source = lambda lo, hi: next(it)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,cfgname,keyset,over", [
    ("tiny", "tiny", "tiny", {}), ("tiny_ls", "tiny", "tiny_learn_sigma", dict(learn_sigma=True)), ("mid", "mid", "tiny", {}),
    # the shipped base model at full size (round 5): the fused VideoConv / temporal-attention / head kernels only run at these shapes
    ("full", "full", "full", {})])
def test_forward_matches_reference(dt, tag, cfgname, keyset, over):
    g = gold(tag + "_forward")
    fl, model, _ = build(cfgname, keyset, dt, **over)
    video, audio = inputs(fl, int(g["B"]), int(g["seed"]))
    replay(model, g["shifts"])
    with torch.no_grad():
        vo, ao = model(video.cuda(), audio.cuda(), torch.from_numpy(g["t"]).cuda())
    assert vo.dtype == torch.float32 and tuple(vo.shape) == g["video_out"].shape
    ev, ea = rel_l2(vo.cpu(), g["video_out"]), rel_l2(ao.cpu(), g["audio_out"])
    print(f"{tag} {dt}: rel-L2 video {ev:.3e} audio {ea:.3e}")
    assert ev < FWD_TOL[dt] and ea < FWD_TOL[dt]


def test_forward_is_repeatable_and_timestep_dtypes_agree():
    fl, model, _ = build("tiny", "tiny", torch.float32)
    video, audio = inputs(fl, 2, 3)
    outs = []
    for t in (torch.tensor([5, 900]), torch.tensor([5, 900], dtype=torch.int32), torch.tensor([5.0, 900.0])):
        model.shift_source = lambda lo, hi: 1 if hi >= 1 else 0
        with torch.no_grad():
            outs.append(model(video.cuda(), audio.cuda(), t.cuda()))
    for vo, ao in outs[1:]:
        assert torch.equal(vo, outs[0][0]) and torch.equal(ao, outs[0][1])       # bitwise repeatable


def test_different_shift_changes_output():
    fl, model, _ = build("tiny", "tiny", torch.float32)
    video, audio = inputs(fl, 1, 4)
    res = []
    for s in (0, 3):
        model.shift_source = lambda lo, hi, s=s: min(s, hi)
        with torch.no_grad():
            res.append(model(video.cuda(), audio.cuda(), torch.tensor([100]).cuda())[0])
    assert rel_l2(res[0].cpu(), res[1].cpu()) > 1e-3


def cpu_noise_source():
    return lambda like: torch.randn(like.shape).to(like.device)


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,cfgname,keyset,resp,over", [
    ("tiny_psample2", "tiny", "tiny", "2", {}), ("tiny_psample4", "tiny", "tiny", "4", {}),
    ("tiny_ls_psample2", "tiny", "tiny_learn_sigma", "2", dict(learn_sigma=True))])
def test_psample_loop_matches_reference(use_graph, dt, tag, cfgname, keyset, resp, over):
    g = gold(tag)
    fl, model, diff = build(cfgname, keyset, dt, timestep_respacing=resp, **over)
    assert diff.timestep_map == list(g["timestep_map"])
    B = int(g["B"])
    replay(model, g["shifts"])
    diff.noise_source = cpu_noise_source()
    torch.manual_seed(int(g["seed"]))
    final = None
    for s in diff.p_sample_loop_progressive(model, {"video": (B, *fl["video_size"]), "audio": (B, *fl["audio_size"])},
                                            device=torch.device("cuda"), use_graph=use_graph):
        final = s
    ev, ea = rel_l2(final["video"].cpu(), g["video"]), rel_l2(final["audio"].cpu(), g["audio"])
    print(f"{tag} {dt} graph={use_graph}: rel-L2 video {ev:.3e} audio {ea:.3e}")
    tol = LOOP_TOL[dt] if dt == torch.float32 else LOOP_TOL_BF16[tag]
    assert ev < tol and ea < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_full_config_two_step_matches_reference(dt):
    """BASELINE config[0] shapes at full size: Landscape base model, batch 1, 2-step DDPM."""
    g = gold("full_psample2")
    fl, model, diff = build("full", "full", dt, timestep_respacing="2")
    replay(model, g["shifts"])
    diff.noise_source = cpu_noise_source()
    torch.manual_seed(int(g["seed"]))
    x = diff.p_sample_loop(model, {"video": (1, *fl["video_size"]), "audio": (1, *fl["audio_size"])},
                           device=torch.device("cuda"), progress=False)
    ev, ea = rel_l2(x["video"].cpu(), g["video"]), rel_l2(x["audio"].cpu(), g["audio"])
    print(f"full 2-step {dt}: rel-L2 video {ev:.3e} audio {ea:.3e}")
    assert ev < LOOP_TOL[dt] and ea < LOOP_TOL[dt]


@pytest.mark.parametrize("tag,keyset,over", [("tiny", "tiny", {}), ("tiny_ls", "tiny_learn_sigma", dict(learn_sigma=True))])
def test_training_loss_forward_values(tag, keyset, over):
    """multimodal_training_losses terms (mse, and vb with learned-range variance) vs the reference's values."""
    g = gold(tag + "_train_loss")
    fl, model, diff = build("tiny", keyset, torch.float32, **over)
    B, seed = int(g["B"]), int(g["seed"])
    gen = torch.Generator().manual_seed(seed)
    x0 = {"video": torch.rand(B, *fl["video_size"], generator=gen) * 2 - 1, "audio": torch.rand(B, *fl["audio_size"], generator=gen) * 2 - 1}
    noise = {"video": torch.randn(B, *fl["video_size"], generator=gen), "audio": torch.randn(B, *fl["audio_size"], generator=gen)}
    replay(model, g["shifts_fwd"])
    with torch.no_grad():
        terms = diff.multimodal_training_losses(model, {k: v.cuda() for k, v in x0.items()}, torch.from_numpy(g["t"]).cuda(),
                                                noise={k: v.cuda() for k, v in noise.items()})
    for k in ("loss", "mse_video", "mse_audio") + (("vb_video", "vb_audio") if over else ()):
        np.testing.assert_allclose(terms[k].cpu().numpy(), g[k], rtol=5e-4, atol=1e-6)


def test_batch_sharding_is_exact():
    """Sharding the batch over ranks must not change any sample: each sample's trajectory is independent
    (GroupNorm / attention never mix batch elements), so rows of a batch-4 run equal two batch-2 runs."""
    fl, model, _ = build("tiny", "tiny", torch.float32)
    video, audio = inputs(fl, 4, 9)
    t = torch.tensor([10, 200, 500, 999])
    model.shift_source = lambda lo, hi: min(2, hi)
    with torch.no_grad():
        vo, ao = model(video.cuda(), audio.cuda(), t.cuda())
        parts = [model(video[i:i + 2].cuda(), audio[i:i + 2].cuda(), t[i:i + 2].cuda()) for i in (0, 2)]
    assert torch.equal(vo, torch.cat([p[0] for p in parts])) and torch.equal(ao, torch.cat([p[1] for p in parts]))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_epilogue_statistics_do_not_change_the_forward(monkeypatch, dt):
    """GroupNorm statistics taken from the producer GEMMs' epilogues vs the statistics pass over every tensor (MMD_GN_EPILOGUE=0).
    fp32 mode (epilogue statistics forced with =2): same output to 1e-4 - the mechanism is exact up to fp32 summation order.  bf16 mode
    (the default): the two plans round differently, so they sit a bf16 noise floor apart (each is within 3e-2 of the fp32 reference,
    test_forward_matches_reference) - bound 3e-2."""
    outs = {}
    for flag in ("2" if dt == torch.float32 else "1", "0"):
        monkeypatch.setenv("MMD_GN_EPILOGUE", flag)
        fl, model, _ = build("mid", "tiny", dt)
        video, audio = inputs(fl, 2, 5)
        model.shift_source = lambda lo, hi: min(1, hi)
        with torch.no_grad():
            outs[flag] = model(video.cuda(), audio.cuda(), torch.tensor([7, 800]).cuda())
        eng = next(iter(model._engines.values()))
        names = [e[2] for e in eng.plan if e[0] is not None]
        n_fin = names.count("mmd_gn_finalize_stats")
        assert (n_fin > 0) == (flag != "0")
        if flag != "0":
            n_pass = names.count("mmd_gn_stats")
            print(f"{dt}: {n_fin} norms finalized from epilogue records, {n_pass} by a statistics pass")
            assert dt == torch.float32 or 4 * n_fin > n_pass        # (64- / 192-channel norms of this config: groups are not whole quads)
    a, b = [v for k, v in outs.items() if k != "0"][0], outs["0"]
    ev, ea = rel_l2(a[0].cpu(), b[0].cpu().numpy()), rel_l2(a[1].cpu(), b[1].cpu().numpy())
    print(f"epilogue statistics vs statistics pass ({dt}): rel-L2 video {ev:.2e} audio {ea:.2e}")
    tol = 1e-4 if dt == torch.float32 else 3e-2
    assert ev < tol and ea < tol
