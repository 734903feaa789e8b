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


def _setup(dt=torch.float32, learn_sigma=False):
    from mm_diffusion import logger, multimodal_script_util as msu
    logger.set_quiet(True)
    g = gold("tiny_ls_train_loss" if learn_sigma else "tiny_train_loss")
    fl = flags("tiny", use_fp16=(dt == torch.bfloat16), learn_sigma=learn_sigma)
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("tiny_learn_sigma" if learn_sigma else "tiny"))
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


def test_learned_sigma_training_gradients_match_reference():
    """learn_sigma=True: loss = mse + vb (KL / decoder NLL with the mean detached); loss terms and parameter gradients vs the
    reference's .backward() (tests/golden/tiny_ls_train_loss.npz; t = [3, 977] exercises the KL branch, the decoder-NLL branch
    is covered by the finite-difference test below)."""
    g, fl, model, diff, x0, noise = _setup(learn_sigma=True)
    terms = diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)
    for k in ("loss", "mse_video", "mse_audio", "vb_video", "vb_audio"):
        np.testing.assert_allclose(terms[k].detach().cpu().numpy(), g[k], rtol=5e-4, atol=1e-6)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        e = rel_l2(params[k].grad.cpu(), g["grad." + k])
        print(f"learn_sigma grad {k}: rel-L2 {e:.2e}")
        assert e < 2e-3, k


def test_loss_terms_backward_matches_finite_differences():
    """mmd_loss_terms_bwd vs central differences of mmd_loss_terms on a tiny tensor, for t == 0 (decoder NLL) and t > 0 (KL)."""
    from mm_diffusion import multimodal_script_util as msu, ops
    fl = flags("tiny", learn_sigma=True)
    _, diff = msu.create_model_and_diffusion(**fl)
    tab, _ = diff.device_tables(torch.device("cuda"))
    N, F, C, HW = 2, 1, 2, 8
    g = torch.Generator().manual_seed(3)
    mo = (torch.randn(N, F, 2 * C, HW, generator=g) * 0.5).cuda()
    x0 = (torch.rand(N, F, C, HW, generator=g) * 2 - 1).cuda()
    x0[0, 0, 0, 0], x0[0, 0, 0, 1] = -1.0, 1.0                       # the two open-ended bins of the discretized likelihood
    eps = torch.randn(N, F, C, HW, generator=g).cuda()
    t = torch.tensor([0, 500], device="cuda")
    _, qtab = diff.device_tables(torch.device("cuda"))
    xt = torch.empty_like(x0)
    ops.q_sample(x0.contiguous(), eps.contiguous(), xt, qtab, t)
    flags_ = 4
    dmse, dvb = torch.tensor([0.7, -0.3], device="cuda"), torch.tensor([1.3, 0.4], device="cuda")

    def total(m):
        mse, vb = ops.loss_terms(m.contiguous(), eps, tab, t, F, C, HW, flags_, x0=x0, xt=xt)
        return float((dmse.double() * mse.double() + dvb.double() * vb.double()).sum())
    gk = torch.empty_like(mo)
    ops.loss_terms_bwd(mo, eps, tab, t, F, C, HW, flags_, dmse, dvb, gk, x0=x0, xt=xt)
    num = torch.zeros_like(mo)
    h = 1e-2
    flat = mo.reshape(-1)
    for i in range(flat.numel()):
        old = float(flat[i])
        flat[i] = old + h
        up = total(mo)
        flat[i] = old - h
        dn = total(mo)
        flat[i] = old
        num.reshape(-1)[i] = (up - dn) / (2 * h)
    # the vb term treats the mean as a constant: compare its gradient on the variance channels, the mse gradient on the mean channels
    np.testing.assert_allclose(gk[:, :, C:].cpu().numpy(), num[:, :, C:].cpu().numpy(), rtol=3e-2, atol=2e-3)
    mse_only = torch.empty_like(mo)
    ops.loss_terms_bwd(mo, eps, tab, t, F, C, HW, flags_, dmse, torch.zeros_like(dvb), mse_only, x0=x0, xt=xt)
    np.testing.assert_allclose(gk[:, :, :C].cpu().numpy(), mse_only[:, :, :C].cpu().numpy(), rtol=1e-6)


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
    # the no-grad engine must notice the update (its packed weights are keyed on the parameters' version counters)
    model.eval()
    model.shift_source = None                 # the recorded draws are used up: fall back to random.randint
    with torch.no_grad():
        model(x0["video"], x0["audio"], torch.from_numpy(g["t"]).cuda())
    eng = next(iter(model._engines.values()))
    opt.zero_grad()
    model.train()
    diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)["loss"].mean().backward()
    opt.step()
    assert eng.stale(), "optimizer step must invalidate the inference engine's packed weights"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_direct_grad_accumulation_and_packed_weights_match_autograd_path(dt):
    """With FlatAdamW the backward kernels accumulate straight into the flat .grad buffer (wgrad in the parameter's own layout)
    and the convs read the one-launch packed weights; the result must equal the plain autograd path (fresh .grad tensors,
    per-call packing) up to the fp32 atomic-add order."""
    import random
    from mm_diffusion.optim import FlatAdamW
    g, fl, model, diff, x0, noise = _setup(dt)
    t = torch.from_numpy(g["t"]).cuda()

    def run():
        random.seed(5)
        model.shift_source = None
        return diff.multimodal_training_losses(model, x0, t, noise=noise)["loss"].mean()
    run().backward()                                   # plain path: autograd allocates the .grad tensors
    ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    opt = FlatAdamW(model.parameters(), lr=1e-4, pack_dtype=model.dtype)
    assert all(hasattr(p, "_mmd_packed") for p in model.parameters() if p.dim() >= 3)
    opt.zero_grad()
    run().backward()
    opt.fold_grads()                                   # conv-weight gradients accumulate in the packed layout until folded
    worst = 0.0
    for k, p in model.named_parameters():
        assert p.grad.data_ptr() >= opt.grad.data_ptr() and p.grad.data_ptr() < opt.grad.data_ptr() + opt.grad.numel() * 4
        worst = max(worst, rel_l2(p.grad.cpu(), ref[k].cpu().numpy()))
    print(f"direct accumulation vs autograd path ({dt}): worst rel-L2 {worst:.2e}")
    assert worst < (1e-5 if dt == torch.float32 else 2e-2)
    # packed copies follow the parameters after a step
    w = dict(model.named_parameters())["middle_blocks.1.v_qkv.weight"]
    opt.step()
    torch.cuda.synchronize()
    Cout, Cin = w.shape[0], w.shape[1]
    assert torch.equal(w._mmd_packed[0], w.detach().reshape(Cout, Cin, -1).permute(0, 2, 1).reshape(Cout, -1).to(dt))
    assert torch.equal(w._mmd_packed[1], w.detach().reshape(Cout, Cin, -1).permute(1, 2, 0).reshape(Cin, -1).to(dt))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_graph_captured_step_matches_eager_step(dt):
    """train_graph.GraphedTrainStep (forward + backward replayed from one captured graph, shifts in device slots) produces the
    same losses and gradients as the eager step given the same batch, timesteps, noise and shift draws."""
    import random
    from mm_diffusion.optim import FlatAdamW
    from mm_diffusion.train_graph import GraphedTrainStep
    g, fl, model, diff, x0, noise = _setup(dt)
    t = torch.from_numpy(g["t"]).cuda()
    opt = FlatAdamW(model.parameters(), lr=0.0, pack_dtype=model.dtype)          # lr 0: the parameters stay put between the runs
    model.shift_source = None
    random.seed(7)
    opt.zero_grad()
    ref_losses = diff.multimodal_training_losses(model, x0, t, noise=noise)
    ref_losses["loss"].mean().backward()
    opt.fold_grads()
    opt._folded = False
    ref_grad = opt.grad.clone()
    ref_loss = ref_losses["loss"].detach().cpu().numpy()
    del ref_losses           # drop the eager autograd graph: its AccumulateGrad nodes belong to the default stream and a capture on
    #                          another stream while they are alive crashes hipStreamEndCapture (torch warns about exactly this)
    gstep = GraphedTrainStep(model, diff, opt, x0)
    gstep.step(x0, t, noise=noise)                                               # warm-up + capture + first replay
    for rep in range(2):                                                         # replays with fresh draws reproduce the eager step
        random.seed(7)
        out = gstep.step(x0, t, noise=noise)
        torch.cuda.synchronize()
        e = rel_l2(opt.grad.cpu(), ref_grad.cpu().numpy())
        print(f"graph replay {rep} vs eager ({dt}): grad rel-L2 {e:.2e}")
        assert e < (1e-5 if dt == torch.float32 else 2e-2)
        np.testing.assert_allclose(out["loss"].cpu().numpy(), ref_loss, rtol=1e-5 if dt == torch.float32 else 2e-2)
    random.seed(8)                                                               # different shifts -> different gradient
    gstep.step(x0, t, noise=noise)
    assert rel_l2(opt.grad.cpu(), ref_grad.cpu().numpy()) > 1e-4
    gstep.close()


def test_non_film_training_gradients_match_reference():
    """use_scale_shift_norm=False (ResBlock: h + emb_out, then the plain norm, unet:473-477): loss terms and parameter gradients vs the
    reference's .backward() (tests/golden/tiny_nofilm_train_loss.npz), including two emb_layers parameters - their gradient is the
    per-sample column sum of the block's upstream gradient (mmd_colsum_slices)."""
    from mm_diffusion import logger, multimodal_script_util as msu
    from mm_diffusion.synth import synth_init_
    logger.set_quiet(True)
    g = gold("tiny_nofilm_train_loss")
    fl = flags("tiny", use_scale_shift_norm=False)
    model, diff = msu.create_model_and_diffusion(**fl)
    synth_init_(model)
    model.cuda().train()
    B, seed = int(g["B"]), int(g["seed"])
    gen = torch.Generator().manual_seed(seed)
    x0 = {"video": (torch.rand(B, *fl["video_size"], generator=gen) * 2 - 1).cuda(), "audio": (torch.rand(B, *fl["audio_size"], generator=gen) * 2 - 1).cuda()}
    noise = {"video": torch.randn(B, *fl["video_size"], generator=gen).cuda(), "audio": torch.randn(B, *fl["audio_size"], generator=gen).cuda()}
    it = iter([int(s) for s in list(g["shifts_fwd"]) + list(g["shifts_bwd"])])
    model.shift_source = lambda lo, hi: next(it)
    terms = diff.multimodal_training_losses(model, x0, torch.from_numpy(g["t"]).cuda(), noise=noise)
    for k in ("loss", "mse_video", "mse_audio"):
        np.testing.assert_allclose(terms[k].detach().cpu().numpy(), g[k], rtol=5e-4)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    for k in GRAD_KEYS + ("input_blocks.1.0.emb_layers.1.weight", "middle_blocks.0.emb_layers.1.bias"):
        e = rel_l2(params[k].grad.cpu(), g["grad." + k])
        print(f"non-FiLM grad {k}: rel-L2 {e:.2e}")
        assert e < 2e-3, k
