"""Timestep samplers for training: uniform, and loss-second-moment importance sampling (what
`create_named_schedule_sampler` of the reference's mm_diffusion/resample.py:8-154 provides to multimodal_train.py).

Contract kept: `sampler.sample(batch_size, device) -> (t int64 [B], w fp32 [B])` with t ~ p and w = 1 / (T p_t), drawn with
`np.random.choice` (so a seeded run picks the same timesteps); `LossAwareSampler.update_with_local_losses(t, losses)` makes
every rank see every rank's (t, loss) pairs; the loss-second-moment weights are sqrt(mean of the last `history_per_term`
squared losses per timestep), mixed with `uniform_prob` of uniform mass, and stay uniform until every timestep has a full
history.

Own structure: the per-timestep history is a ring buffer (the mean of squares does not depend on the order, so overwriting
the oldest entry equals the reference's shift-left), updates are applied per unique timestep with numpy, and the cross-rank
exchange is one size all_gather plus ONE all_gather of a packed [max_bs, 2] fp64 tensor (the reference: three collectives and a
`.item()` per element).
"""
import numpy as np
import torch as th
import torch.distributed as dist


class ScheduleSampler:
    """Base: subclasses provide `weights()` (positive, one per diffusion step, not necessarily normalised)."""

    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = np.asarray(self.weights(), dtype=np.float64)
        p = w / w.sum()
        picked = np.random.choice(p.shape[0], size=(batch_size,), p=p)
        importance = 1.0 / (p.shape[0] * p[picked])
        return th.from_numpy(picked).long().to(device), th.from_numpy(importance).float().to(device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Pool the (t, loss) pairs of all ranks (batch sizes may differ per rank) and update identically everywhere."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        ts = local_ts.detach().reshape(-1)
        ls = local_losses.detach().reshape(-1)
        if world > 1:
            counts = [th.zeros(1, dtype=th.int32, device=ts.device) for _ in range(world)]
            dist.all_gather(counts, th.tensor([ts.numel()], dtype=th.int32, device=ts.device))
            counts = [int(c.item()) for c in counts]
            packed = th.zeros(max(counts), 2, dtype=th.float64, device=ts.device)
            packed[:ts.numel(), 0] = ts.double()
            packed[:ts.numel(), 1] = ls.double()
            everyone = [th.zeros_like(packed) for _ in range(world)]
            dist.all_gather(everyone, packed)
            rows = th.cat([e[:n] for e, n in zip(everyone, counts)]).cpu().numpy()
        else:
            rows = th.stack([ts.double(), ls.double()], dim=1).cpu().numpy()
        self.update_with_all_losses(rows[:, 0].astype(np.int64).tolist(), rows[:, 1].tolist())

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError


class LossSecondMomentResampler(LossAwareSampler):
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        T = diffusion.num_timesteps
        self._loss_history = np.zeros([T, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([T], dtype=np.int64)        # filled entries per timestep (saturates at history_per_term)
        self._cursor = np.zeros([T], dtype=np.int64)             # ring position of the next write

    def _warmed_up(self):
        return bool((self._loss_counts == self.history_per_term).all())

    def weights(self):
        T = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones([T], dtype=np.float64)
        rms = np.sqrt((self._loss_history ** 2).mean(axis=-1))
        rms /= rms.sum()
        return rms * (1 - self.uniform_prob) + self.uniform_prob / T

    def update_with_all_losses(self, ts, losses):
        H = self.history_per_term
        for t, loss in zip(ts, losses):                          # sequential: one batch may hit the same timestep twice
            self._loss_history[t, self._cursor[t]] = loss
            self._cursor[t] = (self._cursor[t] + 1) % H
            if self._loss_counts[t] < H:
                self._loss_counts[t] += 1


def create_named_schedule_sampler(name, diffusion):
    samplers = {"uniform": UniformSampler, "loss-second-moment": LossSecondMomentResampler}
    if name not in samplers:
        raise NotImplementedError(f"unknown schedule sampler: {name}")
    return samplers[name](diffusion)
