"""Training loop for the multimodal U-Net on the MI355X HIP path (reference TrainLoop,
mm_diffusion/multimodal_train_util.py:28-330,470-549): microbatched forward/backward of
`diffusion.multimodal_training_losses`, AdamW, EMA copies, linear lr anneal, loss-aware timestep resampling,
`modelNNNNNN.pt` / `ema_<rate>_NNNNNN.pt` / `optNNNNNN.pt` checkpoints interchangeable with the reference's
(same state-dict keys and tensor shapes) and resume from the newest checkpoint in the log directory.

MI355X-first differences:
  * parameters, gradients and both Adam moments live in ONE flat fp32 buffer each (optim.FlatAdamW): the optimizer + first
    EMA update is one kernel, the data-parallel gradient reduction is ONE RCCL all-reduce of the flat gradient after the
    last microbatch (that is also the reference's `no_sync()` behaviour, mtu:300-305), the initial parameter sync one
    broadcast (the reference: DDP with 128 MB buckets + 1046 per-tensor broadcasts)
  * `use_fp16` selects bf16 activations / conv weights with fp32 master parameters: bf16 has fp32's exponent range, so
    the reference's dynamic loss scaling (fp16_util.py:149-245) has nothing to do and `took_step` is always true
  * the periodic sample dump (`save_video`, mtu:348-467) samples from the first EMA copy WITHOUT touching the master parameters
    (the reference loads the EMA weights into the model), gathers every rank's samples with one RCCL all-gather per stream and
    writes a png grid of frame strips + wav files (no gif / mp4 muxer in this build)
Out of scope here (out of the hot path): wandb logging.
"""
import glob
import os

import torch as th
import torch.distributed as dist

from . import dist_util, logger
from .optim import FlatAdamW
from .resample import LossAwareSampler, UniformSampler


class TrainLoop:
    def __init__(self, *, model, diffusion, data, batch_size, microbatch, ema_rate, log_interval, save_interval, resume_checkpoint,
                 lr=0, t_lr=1e-4, save_type="mp4", use_fp16=False, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.0,
                 lr_anneal_steps=0, class_cond=False, use_db=False, sample_fn="dpm_solver", num_classes=0, save_row=2, video_fps=16,
                 audio_fps=16000, use_graph=False):
        """use_graph (extension): replay forward + backward from one captured graph (train_graph.GraphedTrainStep) when a step is
        a single microbatch of constant shape; ~10 % faster at per-GPU batch 8, identical gradients."""
        self.model, self.diffusion, self.data = model, diffusion, data
        self.save_type = save_type
        self.batch_size = batch_size
        self.microbatch = microbatch if microbatch > 0 else batch_size
        self.lr, self.t_lr = lr, t_lr
        self.ema_rate = [ema_rate] if isinstance(ema_rate, float) else [float(x) for x in ema_rate.split(",")]
        self.log_interval, self.save_interval = log_interval, save_interval
        self.resume_checkpoint = resume_checkpoint
        self.use_fp16, self.fp16_scale_growth = use_fp16, fp16_scale_growth
        self.schedule_sampler = schedule_sampler or UniformSampler(diffusion)
        self.weight_decay, self.lr_anneal_steps = weight_decay, lr_anneal_steps
        self.class_cond, self.num_classes, self.save_row = class_cond, num_classes, save_row
        self.step, self.resume_step = 1, 0
        self.global_batch = self.batch_size * dist_util.world_size()
        self.video_fps, self.audio_fps = video_fps, audio_fps
        self.sample_fn = sample_fn
        self.use_graph = bool(use_graph) and self.microbatch >= self.batch_size
        self._gstep = None
        if use_db:
            raise NotImplementedError("wandb logging (use_db) is not built")
        self._load_and_sync_parameters()
        self.opt = FlatAdamW(self.model.parameters(), lr=self.lr, weight_decay=self.weight_decay, ema_rates=self.ema_rate,
                             pack_dtype=getattr(self.model, "dtype", None))
        self._names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        if self.resume_step:
            self._load_optimizer_state()
            for i, rate in enumerate(self.ema_rate):
                self._load_ema_parameters(i, rate)
        self.output_model_stastics()

    # ------------------------------------------------------------------ state
    def output_model_stastics(self):
        total = sum(p.numel() for p in self.model.parameters())
        train = sum(p.numel() for p in self.opt.params) if self.lr > 0 else 0
        unit, div = ("M", 1e6) if total > 1e6 else (("k", 1e3) if total > 1e3 else ("", 1.0))
        logger.log("Total Parameters:{:.2f}{}".format(total / div, unit))
        logger.log("Total Training Parameters:{:.2f}{}".format(train / div, unit))

    def _load_and_sync_parameters(self):
        resume_checkpoint = find_resume_checkpoint() or self.resume_checkpoint
        if resume_checkpoint:
            self.resume_step = parse_resume_step_from_filename(resume_checkpoint)
            if self.resume_step > 0 and dist_util.rank() == 0:
                logger.log(f"continue training from step {self.resume_step}")
            logger.log(f"loading model from checkpoint: {resume_checkpoint}...")
            self.model.load_state_dict_(dist_util.load_state_dict(resume_checkpoint, map_location=dist_util.dev()))
        dist_util.sync_params(self.model.parameters())

    def _flat_from_state_dict(self, sd, flat):
        off = 0
        for n, p in zip(self._names, self.opt.params):
            flat[off:off + p.numel()].copy_(sd[n].reshape(-1).float())
            off += p.numel()

    def _state_dict_from_flat(self, flat):
        """Reference-compatible state dict (model.state_dict() keys) with the trainable entries read from `flat`."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        off = 0
        for n, p in zip(self._names, self.opt.params):
            sd[n] = flat[off:off + p.numel()].view_as(p).detach().clone().to(sd[n].dtype)
            off += p.numel()
        return sd

    def _load_ema_parameters(self, idx, rate):
        main_checkpoint = find_resume_checkpoint() or self.resume_checkpoint
        ema_checkpoint = find_ema_checkpoint(main_checkpoint, self.resume_step, rate)
        if ema_checkpoint:
            logger.log(f"loading EMA from checkpoint: {ema_checkpoint}...")
            self._flat_from_state_dict(dist_util.load_state_dict(ema_checkpoint, map_location=dist_util.dev()), self.opt.ema_params[idx])
        if dist_util.world_size() > 1:
            dist.broadcast(self.opt.ema_params[idx], 0)

    def _load_optimizer_state(self):
        main_checkpoint = find_resume_checkpoint() or self.resume_checkpoint
        opt_checkpoint = os.path.join(os.path.dirname(main_checkpoint), f"opt{self.resume_step:06}.pt")
        if os.path.exists(opt_checkpoint):
            logger.log(f"loading optimizer state from checkpoint: {opt_checkpoint}")
            self.load_opt_state_dict(dist_util.load_state_dict(opt_checkpoint, map_location=dist_util.dev()))

    def opt_state_dict(self):
        """th.optim.AdamW-shaped state dict ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}) so the
        reference's `opt.load_state_dict` accepts it (mtu:207-220); parameter order = model.parameters()."""
        state, off = {}, 0
        for i, p in enumerate(self.opt.params):
            k = p.numel()
            state[i] = {"step": th.tensor(float(self.opt.steps)), "exp_avg": self.opt.m[off:off + k].view_as(p).clone(),
                        "exp_avg_sq": self.opt.v[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": self.opt.lr, "betas": self.opt.betas, "eps": self.opt.eps, "weight_decay": self.opt.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.opt.params)))}
        return {"state": state, "param_groups": [group]}

    def load_opt_state_dict(self, sd):
        off = 0
        for i, p in enumerate(self.opt.params):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.opt.m[off:off + k].copy_(st["exp_avg"].reshape(-1).float())
                self.opt.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1).float())
                self.opt.steps = int(st["step"])
            off += k

    # ------------------------------------------------------------------ loop
    def run_loop(self):
        while not self.lr_anneal_steps or self.step + self.resume_step < self.lr_anneal_steps:
            batch = next(self.data)
            self.run_step(batch)
            if self.step % self.log_interval == 0:
                logger.dumpkvs()
            if self.step % self.save_interval == 0:
                self.save()
                self.save_video()
                if os.environ.get("DIFFUSION_TRAINING_TEST", "") and self.step > 0:
                    return
            self.step += 1
        if (self.step - 1) % self.save_interval != 0:       # save the last checkpoint if it wasn't already saved
            self.save()

    def _run_step_graph(self, batch):
        from .train_graph import GraphedTrainStep
        batch = {k: v.to(dist_util.dev()) for k, v in batch.items()}
        if self._gstep is None:
            self._gstep = GraphedTrainStep(self.model, self.diffusion, self.opt, batch)
        t, weights = self.schedule_sampler.sample(batch["video"].shape[0], dist_util.dev())
        losses = self._gstep.step(batch, t, weights)          # zero_grad + forward + backward replayed, then all-reduce + AdamW/EMA
        if isinstance(self.schedule_sampler, LossAwareSampler):
            self.schedule_sampler.update_with_local_losses(t, losses["loss"])
        log_loss_dict(self.diffusion, t, {k: v * weights for k, v in losses.items()})
        self._anneal_lr()
        self.log_step()
        return losses

    def run_step(self, batch, cond={}):
        if self.use_graph and not cond:
            return self._run_step_graph(batch)
        self.opt.zero_grad()
        loss = self.forward_backward(batch, cond)
        self.opt.all_reduce_grads()                           # buckets already launched under the last backward are only awaited
        self.opt.step()                                       # AdamW + every EMA copy
        self._anneal_lr()
        self.log_step()
        return loss

    def forward_backward(self, batch, cond):
        batch = {k: v.to(dist_util.dev()) for k, v in batch.items()}
        cond = {k: v.to(dist_util.dev()) for k, v in cond.items()}
        batch_len = batch["video"].shape[0]
        for i in range(0, batch_len, self.microbatch):
            micro = {k: v[i:i + self.microbatch] for k, v in batch.items()}
            micro_cond = {k: v[i:i + self.microbatch] for k, v in cond.items()}
            t, weights = self.schedule_sampler.sample(micro["video"].shape[0], dist_util.dev())
            losses = self.diffusion.multimodal_training_losses(self.model, micro, t, model_kwargs=micro_cond)
            loss = (losses["loss"] * weights).mean()
            if i + self.microbatch >= batch_len:              # last microbatch: gradient buckets are reduced while it is still running
                self.opt.arm_overlap()                        # (the reference: DDP buckets + no_sync() on the earlier ones, mtu:289-319)
            loss.backward()                                   # accumulates straight into the flat gradient buffer
        if isinstance(self.schedule_sampler, LossAwareSampler):
            self.schedule_sampler.update_with_local_losses(t, losses["loss"].detach())
        log_loss_dict(self.diffusion, t, {k: v * weights for k, v in losses.items()})
        return losses

    def _anneal_lr(self):
        if not self.lr_anneal_steps:
            return
        frac_done = (self.step + self.resume_step) / self.lr_anneal_steps
        self.opt.lr = self.lr * (1 - frac_done)

    def log_step(self):
        logger.logkv("step", self.step + self.resume_step)
        logger.logkv("samples", (self.step + self.resume_step + 1) * self.global_batch)

    def save_video(self):
        """Periodic sample dump (mtu:348-467): save_row^2 video+audio pairs from the first EMA copy with the configured sampler
        (`sample_fn`: dpm_solver / dpm_solver++ adaptive-20 like the reference, ddim, or the DDPM loop), every rank's batch gathered
        by one all-gather per stream (mtu:420-431), rank 0 writes `<sample_fn>_samples_steps<N>.png` (grid of frame strips) and one
        wav per sample.  The master parameters are swapped out and back, never overwritten."""
        from .common import save_audio, save_one_video
        dev = dist_util.dev()
        logger.log("create samples...")
        was_training = self.model.training
        keep = None
        if self.opt.ema_params:                               # sample from the EMA weights, then restore the masters
            keep = self.opt.flat.clone()
            self.opt.flat.copy_(self.opt.ema_params[0])
            self._params_changed()
        self.model.eval()
        videos, audios = [], []
        try:
            while len(videos) * self.batch_size * dist_util.world_size() < self.save_row ** 2:
                shape = {"video": [self.batch_size, *self.model.video_size], "audio": [self.batch_size, *self.model.audio_size]}
                with th.no_grad():
                    if self.sample_fn in ("dpm_solver", "dpm_solver++"):
                        from .multimodal_dpm_solver_plus import DPM_Solver
                        pp = self.sample_fn == "dpm_solver++"
                        solver = DPM_Solver(model=self.model, alphas_cumprod=th.tensor(self.diffusion.alphas_cumprod, dtype=th.float32),
                                            predict_x0=pp, thresholding=pp)
                        x_T = {k: th.randn(*v).to(dev) for k, v in shape.items()}
                        sample = solver.sample(x_T, steps=20, order=2, skip_type="logSNR", method="adaptive")
                    else:
                        fn = self.diffusion.ddim_sample_loop if self.sample_fn == "ddim" else self.diffusion.p_sample_loop
                        sample = fn(self.model, shape=shape, clip_denoised=True, model_kwargs={}, device=dev, progress=False)
                videos.append(dist_util.all_gather_samples(sample["video"].float().contiguous()).cpu())
                audios.append(dist_util.all_gather_samples(sample["audio"].float().contiguous()).cpu())
        finally:
            # the sampling engines (activation pools, graphs, a packed copy of the EMA weights) must not stay resident next to the
            # training step and its graph mempool: drop them before the master weights come back
            rel = getattr(self.model, "release_engines", None)
            if rel is not None and th.cuda.is_available():
                th.cuda.synchronize()
                rel()
            if keep is not None:
                self.opt.flat.copy_(keep)
                self._params_changed()
            self.model.train(was_training)
        videos, audios = th.cat(videos), th.cat(audios)
        path = os.path.join(logger.get_dir(), f"{self.sample_fn}_samples_steps{self.step}.png")
        if dist_util.rank() == 0:
            path = save_one_video(videos, path, row=self.save_row)
            for i, a in enumerate(audios[: self.save_row ** 2]):
                save_audio(a.numpy(), os.path.join(logger.get_dir(), f"{self.sample_fn}_samples_steps{self.step}_{i}.wav"), self.audio_fps)
            logger.log(f"{videos.shape[0]} has sampled -> {path}")
        if dist.is_initialized():
            dist.barrier()
        return path

    def _params_changed(self):
        """The flat buffer was rewritten under the parameters: re-pack the GEMM operands and invalidate the inference engines."""
        if self.opt.packer is not None:
            self.opt.packer.refresh()
        bump = getattr(th._C, "_increment_version", None)
        if bump is not None:
            bump(self.opt.params)

    def save(self):
        step = self.step + self.resume_step
        if dist_util.rank() == 0:
            for rate, flat in [(0, self.opt.flat)] + list(zip(self.ema_rate, self.opt.ema_params)):
                logger.log(f"saving model {rate}...")
                filename = f"model{step:06d}.pt" if not rate else f"ema_{rate}_{step:06d}.pt"
                th.save(self._state_dict_from_flat(flat), os.path.join(get_blob_logdir(), filename))
            th.save(self.opt_state_dict(), os.path.join(get_blob_logdir(), f"opt{step:06d}.pt"))
        if dist.is_initialized():
            dist.barrier()


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNN.pt -> NNNNNN (0 if the name has another form)."""
    split = filename.split("model")
    if len(split) < 2:
        return 0
    try:
        return int(split[-1].split(".")[0])
    except ValueError:
        return 0


def get_blob_logdir():
    return logger.get_dir()


def find_resume_checkpoint():
    """Newest modelNNNNNN.pt in the log directory (the reference's auto-resume, mtu:497-510)."""
    logdir = get_blob_logdir()
    if not logdir:
        return None
    max_step = 0
    for name in glob.glob(os.path.join(logdir, "model*.pt")):
        try:
            max_step = max(max_step, int(name[-9:-3]))
        except ValueError:
            pass
    if max_step:
        path = os.path.join(logdir, f"model{max_step:06d}.pt")
        if os.path.exists(path):
            return path
    return None


def find_ema_checkpoint(main_checkpoint, step, rate):
    if main_checkpoint is None:
        return None
    path = os.path.join(os.path.dirname(main_checkpoint), f"ema_{rate}_{step:06d}.pt")
    return path if os.path.exists(path) else None


def log_loss_dict(diffusion, ts, losses):
    """Mean of every loss term plus its mean per timestep quartile (mtu:542-549)."""
    for key, values in losses.items():
        logger.logkv_mean(key, values.mean().item())
        for sub_t, sub_loss in zip(ts.cpu().numpy(), values.detach().cpu().numpy()):
            quartile = int(4 * sub_t / diffusion.num_timesteps)
            logger.logkv_mean(f"{key}_q{quartile}", sub_loss)
