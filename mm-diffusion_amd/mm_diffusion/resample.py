"""Timestep samplers for training (reference mm_diffusion/resample.py:8-154): uniform and loss-second-moment
importance sampling.  Host-side numpy bookkeeping (the reference's is too); only the sampled indices and their
importance weights go to the device.  Differences: `np.int` (removed from numpy) -> np.int64; the per-rank loss
exchange is two fixed-size all_gathers of one packed [max_bs, 2] tensor instead of three collectives and one
`.item()` per element."""
from abc import ABC, abstractmethod

import numpy as np
import torch as th
import torch.distributed as dist


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler(ABC):
    @abstractmethod
    def weights(self):
        """one positive weight per diffusion step (need not be normalised)"""

    def sample(self, batch_size, device):
        """-> (timesteps int64 [B], importance weights fp32 [B] = 1 / (T p_t)), drawn with np.random.choice like the reference."""
        w = self.weights()
        p = w / np.sum(w)
        indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = th.from_numpy(indices_np).long().to(device)
        weights = th.from_numpy(1 / (len(p) * p[indices_np])).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Share this rank's (t, loss) pairs with every rank, then update the reweighting identically everywhere."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1:
            self.update_with_all_losses(local_ts.tolist(), local_losses.tolist())
            return
        sizes = [th.zeros(1, dtype=th.int32, device=local_ts.device) for _ in range(world)]
        dist.all_gather(sizes, th.tensor([len(local_ts)], dtype=th.int32, device=local_ts.device))
        sizes = [int(x.item()) for x in sizes]
        max_bs = max(sizes)
        packed = th.zeros(max_bs, 2, dtype=th.float64, device=local_ts.device)
        packed[:len(local_ts), 0] = local_ts.double()
        packed[:len(local_ts), 1] = local_losses.double()
        gathered = [th.zeros_like(packed) for _ in range(world)]
        dist.all_gather(gathered, packed)
        ts, losses = [], []
        for g, bs in zip(gathered, sizes):
            g = g[:bs].cpu()
            ts += [int(v) for v in g[:, 0].tolist()]
            losses += g[:, 1].tolist()
        self.update_with_all_losses(ts, losses)

    @abstractmethod
    def update_with_all_losses(self, ts, losses):
        """ts: list of int timesteps, losses: list of float losses (identical on every rank)"""


class LossSecondMomentResampler(LossAwareSampler):
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=np.int64)

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        weights = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        weights /= np.sum(weights)
        weights *= 1 - self.uniform_prob
        weights += self.uniform_prob / len(weights)
        return weights

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            if self._loss_counts[t] == self.history_per_term:
                self._loss_history[t, :-1] = self._loss_history[t, 1:]      # shift out the oldest term
                self._loss_history[t, -1] = loss
            else:
                self._loss_history[t, self._loss_counts[t]] = loss
                self._loss_counts[t] += 1

    def _warmed_up(self):
        return (self._loss_counts == self.history_per_term).all()
