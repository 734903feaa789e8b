"""Import-surface counterpart of the reference's FVD / KVD / FAD evaluator (out of the hot path).

`py_scripts/multimodal_sample_sr.py:27` (and the zero-shot scripts) import `eval_multimodal` at module top and call it only when a
reference-set path is given (`sample_sr.py:268`), AFTER the whole sampling loop; `py_scripts/eval.py:24` calls it as its only job.
The metric networks (I3D for FVD / KVD, AudioCLIP for FAD) and the media decoding they need are not part of the denoising path this
package replaces.  What the call does depends on what is at stake:

  * after a sampling loop (any script but eval.py): the finished samples must not be lost to an exception at the very end - the
    call logs at ERROR level, warns, and returns the sentinel {"unavailable": True} (never an empty dict that reads like "no metrics
    were requested"); `warn_metrics_unavailable` says so EARLY, when the flags are parsed;
  * a metrics-only run (the entry script is eval.py) has nothing else to deliver: the call raises EvaluatorUnavailable, so an automated
    metrics pipeline fails instead of logging `metric: {...}` and exiting 0.

MMD_EVAL_STRICT=1 / =0 forces the one or the other behaviour."""
import os
import sys
import warnings


class EvaluatorUnavailable(RuntimeError):
    pass


_warned = set()


def unavailable_message(real_path, fake_path=None):
    return ("mm_diffusion.evaluator.eval_multimodal: the FVD / KVD / FAD evaluator (I3D + AudioCLIP networks) is out of scope for the "
            f"MI355X hot-path package; compute the metrics with the REFERENCE repository's own evaluator on ref_dir={real_path!r}"
            + (f" fake_dir={fake_path!r}" if fake_path is not None else "") + " (sampling itself does not need it)")


def _strict():
    env = os.environ.get("MMD_EVAL_STRICT")
    if env is not None and env != "":
        return env != "0"
    return os.path.basename(sys.argv[0] if sys.argv else "") == "eval.py"      # a metrics-only run


def warn_metrics_unavailable(ref_path):
    """The early notice of a sampling run that was given a reference-set path (called once per path when the flags are parsed)."""
    if not ref_path or ref_path in _warned:
        return
    _warned.add(ref_path)
    msg = unavailable_message(ref_path) + "; the samples will be written, the metrics skipped"
    try:
        from . import logger
        logger.error(msg)
    except Exception:
        pass
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def eval_multimodal(real_path, fake_path, video_size=(16, 3, 64, 64), eval_num=2048):
    """Same signature as the reference (evaluator.py:60).  Raises EvaluatorUnavailable in a metrics-only run (or MMD_EVAL_STRICT=1);
    otherwise logs an error, warns and returns {"unavailable": True} (see the module docstring)."""
    msg = unavailable_message(real_path, fake_path)
    if _strict():
        raise EvaluatorUnavailable(msg)
    try:
        from . import logger
        logger.error(msg + "; returning no metrics")
    except Exception:
        pass
    warnings.warn(msg + "; returning no metrics", RuntimeWarning, stacklevel=2)
    return {"unavailable": True}
