"""End-to-end training step on the HIP path: loss values and parameter gradients of multimodal_training_losses vs the
reference's (tests/golden/tiny_train_loss.npz), including the reference's recompute quirk (the shifted cross-attention
blocks re-draw their window shift when they are re-run in backward, nn.py:262-270), then one AdamW step.

Tolerance: fp32 path, gradients rel-L2 <= 2e-3 against the reference CPU autograd (fp32 sums in a different order over
~1e5-element reductions; measured ~1e-5..1e-4)."""
import numpy as np
import pytest
import torch

from helpers import flags, gold, rel_l2, synth_sd

pytestmark = pytest.mark.gpu

GRAD_KEYS = ("time_embed.0.weight", "input_blocks.0.0.video_conv.video_conv_spatial.weight", "middle_blocks.1.v_qkv.weight",
             "video_out.2.video_conv.bias", "audio_out.2.audio_conv.weight")


def _setup(dt=torch.float32):
    from mm_diffusion import logger, multimodal_script_util as msu
    logger.set_quiet(True)
    g = gold("tiny_train_loss")
    fl = flags("tiny", use_fp16=(dt == torch.bfloat16))
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("tiny"))
    model.cuda().train()
    B, seed = int(g["B"]), int(g["seed"])
    gen = torch.Generator().manual_seed(seed)
    x0 = {"video": torch.rand(B, *fl["video_size"], generator=gen) * 2 - 1, "audio": torch.rand(B, *fl["audio_size"], generator=gen) * 2 - 1}
    noise = {"video": torch.randn(B, *fl["video_size"], generator=gen), "audio": torch.randn(B, *fl["audio_size"], generator=gen)}
    it = iter([int(s) for s in list(g["shifts_fwd"]) + list(g["shifts_bwd"])])
    model.shift_source = lambda lo, hi: next(it)
    return g, fl, model, diff, {k: v.cuda() for k, v in x0.items()}, {k: v.cuda() for k, v in noise.items()}


def test_training_step_gradients_match_reference():
    g, fl, model, diff, x0, noise = _setup()
    terms = diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)
    for k in ("loss", "mse_video", "mse_audio"):
        np.testing.assert_allclose(terms[k].detach().cpu().numpy(), g[k], rtol=5e-4)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        e = rel_l2(params[k].grad.cpu(), g["grad." + k])
        print(f"grad {k}: rel-L2 {e:.2e}")
        assert e < 2e-3, k
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params.values())


def test_bf16_training_step_runs_and_is_close():
    g, fl, model, diff, x0, noise = _setup(torch.bfloat16)
    terms = diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)
    np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), g["loss"], rtol=3e-2)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        e = rel_l2(params[k].grad.cpu(), g["grad." + k])
        print(f"bf16 grad {k}: rel-L2 {e:.2e}")
        assert e < 0.15, k


def test_flat_adamw_step_updates_every_parameter():
    from mm_diffusion.optim import FlatAdamW
    g, fl, model, diff, x0, noise = _setup()
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=0.0, ema_rates=[0.9999])
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)["loss"].mean().backward()
    opt.step()
    moved = [k for k, v in model.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(moved) == len(before)
    # first Adam step moves every element by ~lr (sign of the gradient)
    p = dict(model.named_parameters())["middle_blocks.1.v_qkv.weight"]
    d = (p.detach() - before["middle_blocks.1.v_qkv.weight"]).abs()
    assert float(d.max()) <= 1.01e-4 and float(d.median()) > 0.5e-4
    assert len(opt.ema_params) == 1 and opt.ema_params[0].numel() == sum(v.numel() for v in before.values())
