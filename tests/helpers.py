"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "tiny": dict(video_size=[8, 3, 16, 16], audio_size=[1, 512], num_channels=64,
                 num_head_channels=32, num_res_blocks=1, channel_mult="1,2,3,4",
                 resblock_updown=True),
    "mid": dict(video_size=[16, 3, 32, 32], audio_size=[1, 6400], num_channels=64,
                num_head_channels=32, num_res_blocks=1, channel_mult="1,2,3,4",
                resblock_updown=True),
    "full": dict(video_size=[16, 3, 64, 64], audio_size=[1, 25600], num_channels=128,
                 num_head_channels=64, num_res_blocks=2, resblock_updown=True),
}

# the reference's flag defaults (multimodal_script_util.py:12-55), restated as data
DEFAULT_FLAGS = dict(
    video_size="16,3,64,64", audio_size="1,25600", num_channels=128, num_res_blocks=2, num_heads=4,
    num_heads_upsample=-1, num_head_channels=-1, cross_attention_resolutions="2,4,8",
    cross_attention_windows="1,4,8", cross_attention_shift=True, video_attention_resolutions="2,4,8",
    audio_attention_resolutions="-1", channel_mult="", dropout=0.0, class_cond=False,
    use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False, use_fp16=False,
    video_type="2d+1d", audio_type="1d",
    learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="",
    use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
)


def flags(name, **over):
    d = dict(DEFAULT_FLAGS)
    d.update(CONFIGS[name])
    d.update(over)
    return d


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def gold_keys():
    with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
        return json.load(f)


def synth_sd(name_in_keys_json):
    from mm_diffusion.synth import synth_tensor
    return {k: synth_tensor(k, s) for k, s in gold_keys()[name_in_keys_json]}


def inputs(f, B, seed):
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, *f["video_size"], generator=g)
    audio = torch.randn(B, *f["audio_size"], generator=g)
    return video, audio


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
