"""`load_data` for the training script (reference multimodal_datasets.py:16-120 decodes mp4 clips with torchvision / moviepy:
dataset IO, out of the hot path and not available here).  This counterpart keeps the generator contract -
`yield {"video": [B, F, C, H, W] float in [-1, 1], "audio": [B, C, L] float}` forever, rank-sharded - over PRE-EXTRACTED clips:
every `*.npz` under `data_dir` holding `video` (uint8 [F, H, W, 3] or float [F, 3, H, W]) and `audio` ([L] or [C, L]) arrays.
`data_dir="synthetic"` yields random clips (smoke-training without data)."""
import glob
import os

import numpy as np
import torch as th

from . import dist_util


def _clip(path, video_size, audio_size):
    z = np.load(path)
    v, a = z["video"], z["audio"]
    if v.dtype == np.uint8:
        v = np.transpose(v, (0, 3, 1, 2)).astype(np.float32) / 127.5 - 1
    v = th.from_numpy(np.ascontiguousarray(v, dtype=np.float32))[:video_size[0]]
    a = th.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(audio_size[0], -1)[:, :audio_size[1]]
    if tuple(v.shape) != tuple(video_size) or tuple(a.shape) != tuple(audio_size):
        raise ValueError(f"{path}: clip shapes {tuple(v.shape)} / {tuple(a.shape)} do not match {tuple(video_size)} / {tuple(audio_size)} "
                         "(resize when extracting; this loader does not resample)")
    return v, a


def load_data(*, data_dir, batch_size, video_size, audio_size, deterministic=False, random_flip=True, num_workers=0, video_fps=10,
              audio_fps=None, frame_gap=1, drop_last=True):
    if not data_dir:
        raise ValueError("unspecified data directory")
    if data_dir == "synthetic":
        g = th.Generator().manual_seed(1234 + dist_util.rank())
        while True:
            yield {"video": th.rand(batch_size, *video_size, generator=g) * 2 - 1, "audio": th.rand(batch_size, *audio_size, generator=g) * 2 - 1}
    files = sorted(glob.glob(os.path.join(data_dir, "**", "*.npz"), recursive=True))[dist_util.rank()::dist_util.world_size()]
    if not files:
        raise ValueError(f"no pre-extracted *.npz clips under {data_dir} (raw video decoding is not part of this build)")
    rng = np.random.default_rng(None if not deterministic else 0)
    while True:
        order = np.arange(len(files)) if deterministic else rng.permutation(len(files))
        for i in range(0, len(order) - (batch_size - 1 if drop_last else 0), batch_size):
            clips = [_clip(files[j], video_size, audio_size) for j in order[i:i + batch_size]]
            v = th.stack([c[0] for c in clips])
            if random_flip and not deterministic:
                flip = th.from_numpy(rng.random(v.shape[0]) < 0.5)
                v[flip] = v[flip].flip(-1)
            yield {"video": v, "audio": th.stack([c[1] for c in clips])}
