"""Multimodal DPM-Solver / DPM-Solver++ on the HIP path against fixtures captured from the reference's
multimodal_dpm_solver_plus.DPM_Solver.sample (tools/gen_golden.py), same x_T, same replayed window shifts.

Tolerance: rel-L2 <= 1e-4 in fp32 mode over 7-68 network evaluations (measured 3e-7 .. 1.2e-6; the random-weight tiny
model is an expanding map - |x| reaches ~700 from N(0,1) inputs in the noise-prediction solvers).  The adaptive solvers
must also take the SAME accept / reject decisions, which shows up as the same number of consumed shift draws."""
import numpy as np
import pytest
import torch

from helpers import flags, gold, rel_l2
from test_model_gpu import build

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_dpm_singlestep3": (False, False, dict(steps=20, order=3, skip_type="logSNR", method="singlestep")),
    "tiny_dpm_singlestep2": (False, False, dict(steps=7, order=2, skip_type="time_quadratic", method="singlestep")),
    "tiny_dpm_multistep2": (False, False, dict(steps=10, order=2, skip_type="time_uniform", method="multistep")),
    "tiny_dpmpp_multistep2": (True, True, dict(steps=10, order=2, skip_type="logSNR", method="multistep", denoise=True)),
    "tiny_dpmpp_adaptive2": (True, True, dict(steps=20, order=2, skip_type="logSNR", method="adaptive")),
    "tiny_dpm_adaptive3": (False, False, dict(order=3, method="adaptive", atol=0.05, rtol=0.1)),
}


def _run(tag, dt=torch.float32):
    from mm_diffusion.multimodal_dpm_solver_plus import DPM_Solver
    g = gold(tag)
    predict_x0, thresholding, kw = CASES[tag]
    fl, model, diff = build("tiny", "tiny", dt)
    B = int(g["B"])
    used = []
    it = iter(int(s) for s in g["shifts"])

    def src(lo, hi):
        v = next(it)
        used.append(v)
        return v
    model.shift_source = src
    torch.manual_seed(int(g["seed"]))
    x_T = {"video": torch.randn(B, *fl["video_size"]).cuda(), "audio": torch.randn(B, *fl["audio_size"]).cuda()}
    solver = DPM_Solver(model=model, alphas_cumprod=torch.tensor(diff.alphas_cumprod, dtype=torch.float32), predict_x0=predict_x0,
                        thresholding=thresholding)
    out = solver.sample(x_T, **kw)
    return g, out, used, solver


@pytest.mark.parametrize("tag", list(CASES))
def test_dpm_solver_matches_reference(tag):
    g, out, used, solver = _run(tag)
    ev, ea = rel_l2(out["video"].cpu(), g["video"]), rel_l2(out["audio"].cpu(), g["audio"])
    print(f"{tag}: rel-L2 video {ev:.3e} audio {ea:.3e}, {solver.nfe} network evaluations")
    assert len(used) == len(g["shifts"]), "different number of network evaluations than the reference"
    assert ev < 1e-4 and ea < 1e-4


def test_batch_of_one_fails_like_the_reference():
    from mm_diffusion.multimodal_dpm_solver_plus import DPM_Solver
    fl, model, diff = build("tiny", "tiny", torch.float32)
    solver = DPM_Solver(model=model, alphas_cumprod=torch.tensor(diff.alphas_cumprod, dtype=torch.float32))
    x = {"video": torch.randn(1, *fl["video_size"]).cuda(), "audio": torch.randn(1, *fl["audio_size"]).cuda()}
    with pytest.raises(AttributeError):
        solver.sample(x, steps=4, order=2, method="multistep")


def test_abs_quantile_and_threshold_kernels():
    from mm_diffusion import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 8, 3, 16, 16, generator=g) * torch.tensor([0.3, 1.0, 4.0]).view(3, 1, 1, 1, 1)).cuda().contiguous()
    q = ops.abs_quantile(x, 0.995)
    ref = torch.quantile(x.abs().reshape(3, -1).cpu(), 0.995, dim=1)
    np.testing.assert_allclose(q.cpu().numpy(), ref.numpy(), rtol=1e-6)
    y = x.clone()
    ops.clamp_scale_(y, q, 1.0)
    s = torch.maximum(ref, torch.ones_like(ref)).view(3, 1, 1, 1, 1)
    np.testing.assert_allclose(y.cpu().numpy(), (torch.clamp(x.cpu(), -s, s) / (s / 1.0)).numpy(), rtol=1e-6, atol=1e-7)


def test_unbuilt_variants_raise_with_a_reason():
    from mm_diffusion.multimodal_dpm_solver_plus import DPM_Solver
    fl, model, diff = build("tiny", "tiny", torch.float32)
    solver = DPM_Solver(model=model, alphas_cumprod=torch.tensor(diff.alphas_cumprod, dtype=torch.float32))
    x = {"video": torch.randn(2, *fl["video_size"]).cuda(), "audio": torch.randn(2, *fl["audio_size"]).cuda()}
    with pytest.raises(NotImplementedError):
        solver.sample(x, steps=6, order=3, method="multistep")
    with pytest.raises(NotImplementedError):
        solver.sample(x, steps=6, order=3, method="singlestep", solver_type="taylor")
