"""Import-surface counterpart of the reference's FVD / KVD / FAD evaluator (out of the hot path).

`py_scripts/multimodal_sample_sr.py:27` (and the zero-shot scripts) import `eval_multimodal` at module top and call it only when a
reference-set path is given (`sample_sr.py:268`), AFTER the whole sampling loop.  The metric networks (I3D for FVD / KVD, AudioCLIP
for FAD) and the media decoding they need are not part of the denoising path this package replaces.  A run that asks for them must
not lose its samples to an exception at the very end: the call logs what to do instead and returns an empty metric dict (the
script then logs `{}` and finishes); `multimodal_script_util.args_to_dict` already says so when the flags are parsed.
MMD_EVAL_STRICT=1 turns the call into an error (EvaluatorUnavailable) for pipelines that must not continue without metrics."""
import os
import warnings


class EvaluatorUnavailable(RuntimeError):
    pass


def unavailable_message(real_path, fake_path=None):
    return ("mm_diffusion.evaluator.eval_multimodal: the FVD / KVD / FAD evaluator (I3D + AudioCLIP networks) is out of scope for the "
            f"MI355X hot-path package; run the reference's py_scripts/eval.py on ref_dir={real_path!r}"
            + (f" fake_dir={fake_path!r}" if fake_path is not None else "") + " (sampling itself does not need it)")


def eval_multimodal(real_path, fake_path, video_size=(16, 3, 64, 64), eval_num=2048):
    """Same signature as the reference (evaluator.py:60).  Returns {} after a warning (see the module docstring); raises
    EvaluatorUnavailable under MMD_EVAL_STRICT=1."""
    msg = unavailable_message(real_path, fake_path)
    if os.environ.get("MMD_EVAL_STRICT", "0") not in ("", "0"):
        raise EvaluatorUnavailable(msg)
    warnings.warn(msg + "; returning no metrics", RuntimeWarning, stacklevel=2)
    return {}
