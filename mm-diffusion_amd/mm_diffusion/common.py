"""Script-level helpers `py_scripts/*.py` import from `mm_diffusion.common` (reference common.py:20-122): run set-up and sample
writers.  Out of the hot path; kept thin.  Writers use what this environment has (PIL for frames, the stdlib `wave` module for
audio); the reference's mp4 / gif writers need moviepy, which is not a dependency here, so `save_multimodal` / `save_video` write
a frame strip (.png) + a .wav next to the requested path and say so."""
import glob
import os
import wave

import numpy as np
import torch as th

from . import dist_util, logger


def set_seed_logger_random(args):
    """Per-rank randomness is left unseeded on purpose (reference common.py:103-122): every rank must draw different noise."""
    if not os.path.exists(args.output_dir) and dist_util.rank() == 0:
        os.makedirs(args.output_dir, exist_ok=True)
    if dist_util.rank() == 0:
        logger.log("Effective parameters:")
        for key in sorted(args.__dict__):
            logger.log("  <<< {}: {}".format(key, args.__dict__[key]))
    return args


def set_seed_logger(args):
    """Seeded variant (common.py:84-101): python / numpy / torch seeded with args.seed."""
    import random
    random.seed(args.seed)
    os.environ["PYTHONHASHSEED"] = str(args.seed)
    np.random.seed(args.seed)
    th.manual_seed(args.seed)
    if th.cuda.is_available():
        th.cuda.manual_seed_all(args.seed)
    return set_seed_logger_random(args)


def delete_pkl(fake_dir):
    for f in glob.glob(os.path.join(fake_dir, "*.pkl")):
        os.remove(f)


def save_audio(audio, output_path, audio_fps):
    """audio [C, L] float in [-1, 1] -> 16-bit PCM .wav"""
    a = np.asarray(audio, dtype=np.float32)
    a = a.reshape(1, -1) if a.ndim == 1 else a
    pcm = (np.clip(a.T, -1, 1) * 32767).astype("<i2")
    with wave.open(output_path, "wb") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(int(audio_fps))
        w.writeframes(pcm.tobytes())
    return output_path


def save_png(img, output_path):
    from PIL import Image
    Image.fromarray(np.asarray(img, dtype=np.uint8)).save(output_path)
    return output_path


def save_img(video, output_path):
    """video [F, H, W, C] uint8 -> one png per frame under output_path/"""
    os.makedirs(output_path, exist_ok=True)
    for i, frame in enumerate(np.asarray(video)):
        save_png(frame, os.path.join(output_path, f"{i:04d}.png"))
    return output_path


def save_multimodal(video, audio, output_path, args):
    """Frame strip + wav instead of the reference's muxed mp4 (moviepy is not available here)."""
    base = os.path.splitext(output_path)[0]
    v = np.asarray(video, dtype=np.uint8)
    save_png(np.concatenate(list(v), axis=1), base + "_frames.png")
    save_audio(audio, base + ".wav", getattr(args, "audio_fps", 16000))
    logger.log(f"save_multimodal: wrote {base}_frames.png and {base}.wav (no mp4 muxer in this build)")
    return base + "_frames.png"


def save_one_video(videos, save_path, row=5):
    """videos [B, F, C, H, W] float in [-1, 1] -> a png grid of frame strips"""
    v = ((th.as_tensor(videos).float() + 1) * 127.5).clamp(0, 255).to(th.uint8).cpu().numpy()
    rows = [np.concatenate([np.transpose(f, (1, 2, 0)) for f in clip], axis=1) for clip in v[:row * row]]
    path = os.path.splitext(save_path)[0] + ".png"
    save_png(np.concatenate(rows, axis=0), path)
    return path
