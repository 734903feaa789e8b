"""Image super-resolution stage (ImageSuperResModel + tensor-valued diffusion) on the HIP path against fixtures captured
from the reference (tools/gen_golden.py: gen_sr).  Tolerances as the multimodal model: fp32 forward 1e-4 / loops 5e-4,
bf16 forward 3e-2 / loops 1e-1."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, gold, rel_l2

pytestmark = pytest.mark.gpu

TINY = dict(large_size=64, small_size=16, sr_num_channels=32, sr_num_res_blocks=1, sr_attention_resolutions="2,4", sr_num_heads=2,
            sr_resblock_updown=True)


def build(dt, **over):
    from mm_diffusion import logger, script_util as su
    from mm_diffusion.synth import synth_tensor
    logger.set_quiet(True)
    d = su.image_sr_model_and_diffusion_defaults()
    d.update(TINY)
    d.update(use_fp16=(dt == torch.bfloat16), **over)
    model, diff = su.image_sr_create_model_and_diffusion(**d)
    with open(os.path.join(GOLD, "sr_state_dict_keys.json")) as f:
        keys = json.load(f)["tiny"]
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == [(k, s) for k, s in keys]     # keys, shapes AND order
    model.load_state_dict({k: synth_tensor(k, s) for k, s in keys})
    model.cuda().eval()
    return model, diff


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_sr_forward_matches_reference(dt):
    g = gold("sr_tiny_forward")
    model, _ = build(dt)
    with torch.no_grad():
        y = model(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda(), low_res=torch.from_numpy(g["low"]).cuda())
    e = rel_l2(y.cpu(), g["y"])
    print(f"sr forward {dt}: rel-L2 {e:.3e}")
    assert y.shape == (2, 6, 64, 64) and e < (1e-4 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,resp,fn", [("sr_tiny_ddim4", "ddim4", "ddim_sample_loop"), ("sr_tiny_ddpm3", "3", "p_sample_loop")])
def test_sr_sampling_loops_match_reference(dt, tag, resp, fn):
    g = gold(tag)
    model, diff = build(dt, sr_timestep_respacing=resp)
    assert diff.timestep_map == list(g["timestep_map"])
    low, noise = torch.from_numpy(g["low"]).cuda(), torch.from_numpy(g["noise"]).cuda()
    diff.noise_source = lambda like: torch.randn(like.shape).to(like.device)
    torch.manual_seed(72)
    torch.randn(1, 3, 64, 64)            # the fixture drew its start noise from this seed first
    out = getattr(diff, fn)(model, tuple(noise.shape), clip_denoised=True, model_kwargs={"low_res": low}, noise=noise.clone(),
                            device=torch.device("cuda"), progress=False)
    e = rel_l2(out.cpu(), g["sample"])
    print(f"{tag} {dt}: rel-L2 {e:.3e}")
    assert e < (5e-4 if dt == torch.float32 else 1e-1)


def test_bilinear_concat_kernel():
    from mm_diffusion import ops
    g = torch.Generator().manual_seed(9)
    x, low = torch.randn(2, 3, 24, 24, generator=g), torch.randn(2, 3, 7, 5, generator=g)
    out = torch.empty(2, 6, 24, 24, device="cuda")
    ops.bilinear_concat(x.cuda(), low.cuda(), out)
    ref = torch.cat([x, torch.nn.functional.interpolate(low, (24, 24), mode="bilinear")], dim=1)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt,impl,tol", [(torch.bfloat16, 0, 1e-2), (torch.bfloat16, 1, 1e-2), (torch.float32, 0, 2e-6)])
@pytest.mark.parametrize("ch", [192, 160])
def test_attention_head_width_192(dt, impl, tol, ch):
    """The shipped SR model has 768 channels / 4 heads = 192-wide heads at ds 16 / 32 (/root/reference/mm_diffusion/image_unet.py:336-353):
    the bf16 MFMA kernel (impl 0, width 192), the VALU kernel on bf16 rows (impl 1) and the fp32-mode VALU kernel, all vs fp32 torch on
    the same rows.  160: a width only the VALU kernel's half-key-tile form covers."""
    from mm_diffusion import ops
    N, T, heads = 2, 256 + 37, 4
    C = heads * ch
    g = torch.Generator().manual_seed(2)
    qkv = (torch.randn(N * T, 3 * C, generator=g) * 0.5).to(dt)
    out = torch.empty(N * T, C, dtype=dt, device="cuda")
    ops.attn(qkv.cuda(), qkv.cuda(), out, heads, ch, N, 1, T, T, T, T, 1, impl=impl)
    q, k, v = [t.reshape(N, T, heads, ch).permute(0, 2, 1, 3) for t in qkv.double().split(C, dim=1)]
    ref = torch.softmax(q @ k.transpose(-1, -2) / ch ** 0.5, dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(N * T, C).float()
    assert rel_l2(out.float().cpu(), ref) < tol


def test_unbuilt_sr_variants_raise():
    from mm_diffusion import script_util as su
    d = su.image_sr_model_and_diffusion_defaults()
    d.update(TINY)
    d.update(sr_resblock_updown=False)
    with pytest.raises(NotImplementedError):
        su.image_sr_create_model_and_diffusion(**d)


@pytest.mark.parametrize("tag,px0", [("sr_tiny_dpm_multistep2", False), ("sr_tiny_dpmpp_multistep2", True)])
def test_sr_single_modal_dpm_solver_matches_reference(tag, px0):
    """dpm_solver_plus.DPM_Solver on the SR model, called like multimodal_sample_sr.py:199-228."""
    from mm_diffusion.dpm_solver_plus import DPM_Solver
    g = gold(tag)
    model, diff = build(torch.float32)
    low, noise = torch.from_numpy(g["low"]).cuda(), torch.from_numpy(g["noise"]).cuda()
    solver = DPM_Solver(model=model, alphas_cumprod=torch.tensor(diff.alphas_cumprod, dtype=torch.float32), predict_x0=px0,
                        model_kwargs={"low_res": low})
    out = solver.sample(noise.clone(), steps=6, order=2, skip_type="time_uniform", method="multistep")
    e = rel_l2(out.cpu(), g["sample"])
    print(f"{tag}: rel-L2 {e:.3e}")
    assert out.shape == noise.shape and e < 1e-4


def test_sr_graph_replay_equals_eager(monkeypatch):
    """MMD_SR_GRAPH=1: the recorded + captured forward (per-shape tile autotune, static buffers) equals the eager launches bitwise, on the
    first evaluation (capture) and on replays with new inputs."""
    g = gold("sr_tiny_forward")
    model, _ = build(torch.bfloat16)
    x, t, low = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda(), torch.from_numpy(g["low"]).cuda()
    with torch.no_grad():
        eager = [model(x, t, low_res=low), model(x * 0.5, t + 3, low_res=low)]
        monkeypatch.setenv("MMD_SR_GRAPH", "1")
        replay = [model(x, t, low_res=low), model(x * 0.5, t + 3, low_res=low), model(x, t, low_res=low)]
    assert len(model._graphs) == 1
    assert torch.equal(eager[0], replay[0]) and torch.equal(eager[1], replay[1]) and torch.equal(eager[0], replay[2])
