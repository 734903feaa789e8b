"""TrainLoop on the HIP path: steps run, checkpoints carry the reference's names / state-dict keys, and a resumed loop
continues bit-for-bit from the saved model / EMA / optimizer state."""
import os

import numpy as np
import pytest
import torch

from helpers import flags, synth_sd

pytestmark = pytest.mark.gpu


def _mk(tmp, dt=torch.float32, **kw):
    from mm_diffusion import logger, multimodal_script_util as msu
    from mm_diffusion.multimodal_train_util import TrainLoop
    logger.set_quiet(True)
    logger.configure(dir=str(tmp))
    fl = flags("tiny", use_fp16=(dt == torch.bfloat16))
    model, diff = msu.create_model_and_diffusion(**fl)
    model.load_state_dict(synth_sd("tiny"))
    model.cuda().train()

    def data():
        g = torch.Generator().manual_seed(0)
        while True:
            yield {"video": torch.rand(4, *fl["video_size"], generator=g) * 2 - 1, "audio": torch.rand(4, *fl["audio_size"], generator=g) * 2 - 1}
    args = dict(model=model, diffusion=diff, data=data(), batch_size=4, microbatch=2, ema_rate="0.9,0.99", log_interval=1, save_interval=2,
                resume_checkpoint="", lr=1e-4, weight_decay=0.01, lr_anneal_steps=0)
    args.update(kw)
    return model, TrainLoop(**args)


def _seed():
    import random
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)


def test_checkpoints_and_resume(tmp_path):
    _seed()
    model, loop = _mk(tmp_path / "a", lr_anneal_steps=3)       # steps 1, 2 -> save at 2 -> loop ends at step 3
    w0 = loop.opt.flat.clone()
    loop.run_loop()
    assert not torch.equal(w0, loop.opt.flat)
    names = sorted(os.listdir(tmp_path / "a"))
    assert {"model000002.pt", "ema_0.9_000002.pt", "ema_0.99_000002.pt", "opt000002.pt"} <= set(names)
    sd = torch.load(tmp_path / "a" / "model000002.pt")
    assert list(sd.keys()) == list(model.state_dict().keys())                 # reference-compatible keys, same order
    ema = torch.load(tmp_path / "a" / "ema_0.9_000002.pt")
    assert any(not torch.equal(sd[k].cpu(), ema[k].cpu()) for k in sd)
    osd = torch.load(tmp_path / "a" / "opt000002.pt")
    assert set(osd.keys()) == {"state", "param_groups"} and len(osd["state"]) == len(list(model.parameters()))
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p)) for p in model.parameters()])
    ref_opt.load_state_dict(osd)                                               # th.optim.AdamW accepts it (mtu:207-220)

    # resume: a fresh loop in the same log dir picks up step 2 with identical model / EMA / moments
    _seed()
    model2, loop2 = _mk(tmp_path / "a", lr_anneal_steps=5)
    assert loop2.resume_step == 2
    for k, v in model2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu())
    off = 0
    for i, p in enumerate(loop2.opt.params):
        assert torch.equal(loop2.opt.m[off:off + p.numel()].view_as(p).cpu(), osd["state"][i]["exp_avg"].cpu())
        off += p.numel()
    assert loop2.opt.steps == 2
    loop2.run_loop()
    assert "model000004.pt" in os.listdir(tmp_path / "a")


def test_loss_goes_down_bf16(tmp_path):
    """Ten AdamW steps on one repeated batch lower the training loss (bf16 activations, fp32 master parameters)."""
    _seed()
    model, loop = _mk(tmp_path / "b", dt=torch.bfloat16, lr=2e-4, ema_rate="0.999")
    batch = next(loop.data)
    losses = []
    for _ in range(10):
        np.random.seed(1)               # same timesteps every step: the loss is comparable
        import random
        random.seed(1)
        torch.manual_seed(1)
        out = loop.run_step(batch)
        losses.append(float(out["loss"].detach().mean()))
        loop.step += 1
    print("losses", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]


def test_loss_second_moment_sampler_updates():
    from mm_diffusion.resample import create_named_schedule_sampler

    class D:
        num_timesteps = 4
    s = create_named_schedule_sampler("loss-second-moment", D())
    for _ in range(10):
        s.update_with_local_losses(torch.tensor([0, 1, 2, 3]), torch.tensor([1.0, 2.0, 3.0, 4.0]))
    w = s.weights()
    assert w[3] > w[0] and abs(w.sum() - 1) < 1e-9
    t, iw = s.sample(8, torch.device("cpu"))
    assert t.shape == (8,) and iw.shape == (8,)


def test_graph_captured_loop_trains(tmp_path):
    """TrainLoop(use_graph=True): the captured forward+backward step lowers the loss on a repeated batch like the eager loop."""
    import random
    _seed()
    model, loop = _mk(tmp_path / "c", dt=torch.bfloat16, lr=2e-4, ema_rate="0.999", microbatch=4, use_graph=True)
    batch = next(loop.data)
    losses = []
    for _ in range(8):
        np.random.seed(1)
        random.seed(1)
        torch.manual_seed(1)
        out = loop.run_step(batch)
        losses.append(float(out["loss"].mean()))
        loop.step += 1
    print("graph-loop losses", [round(v, 4) for v in losses])
    assert loop._gstep is not None and losses[-1] < losses[0]


@pytest.mark.parametrize("sample_fn", ["dpm_solver++", "ddim"])
def test_periodic_sample_dump_uses_ema_and_restores_the_masters(tmp_path, sample_fn):
    """save_video (mtu:348-467): samples come from the first EMA copy, the master parameters are back afterwards (bitwise), rank 0 writes
    the png grid of frame strips and one wav per sample."""
    from mm_diffusion import logger
    _seed()
    model, loop = _mk(tmp_path / sample_fn, sample_fn=sample_fn, save_row=2)
    for _ in range(2):
        loop.run_step(next(loop.data))
        loop.step += 1
    masters, ema0 = loop.opt.flat.clone(), loop.opt.ema_params[0].clone()
    assert not torch.equal(masters, ema0)
    train_diff = loop.diffusion
    if sample_fn == "ddim":                                   # keep the DDIM loop short: sample on a 4-step respacing
        from mm_diffusion import multimodal_script_util as msu
        loop.diffusion = msu.create_gaussian_diffusion(steps=1000, timestep_respacing="ddim4")
    path = loop.save_video()
    loop.diffusion = train_diff
    assert torch.equal(loop.opt.flat, masters) and torch.equal(loop.opt.ema_params[0], ema0)
    assert os.path.exists(path) and path.endswith(".png")
    wavs = [f for f in os.listdir(logger.get_dir()) if f.endswith(".wav")]
    assert len(wavs) == 4
    from PIL import Image
    im = Image.open(path)
    assert im.size == (8 * 16, 4 * 16)                        # 4 clips (rows) x 8 frames of 16 x 16
    loop.run_step(next(loop.data))                            # training goes on from the master parameters
    assert torch.isfinite(loop.opt.flat).all()
