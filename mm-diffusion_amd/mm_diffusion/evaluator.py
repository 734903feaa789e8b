"""Import-surface counterpart of the reference's FVD / KVD / FAD evaluator (out of the hot path).

`py_scripts/multimodal_sample_sr.py:27` (and the zero-shot scripts) import `eval_multimodal` at module top and call it only when a
reference-set path is given (`sample_sr.py:268`).  The metric networks (I3D for FVD / KVD, AudioCLIP for FAD) and the media decoding
they need are not part of the denoising path this package replaces, so the name resolves and the CALL says what to do instead."""


class EvaluatorUnavailable(RuntimeError):
    pass


def eval_multimodal(real_path, fake_path, video_size=(16, 3, 64, 64), eval_num=2048):
    """Same signature as the reference (evaluator.py:60).  Raises: the FVD / KVD / FAD evaluator is outside the scope of this
    package - score the generated files in `fake_path` with the reference repository's `py_scripts/eval.py`."""
    raise EvaluatorUnavailable(
        "mm_diffusion.evaluator.eval_multimodal: the FVD / KVD / FAD evaluator (I3D + AudioCLIP networks) is out of scope for the "
        f"MI355X hot-path package; run the reference's py_scripts/eval.py on ref_dir={real_path!r} fake_dir={fake_path!r} "
        "(sampling itself does not need it: leave --ref_path empty)")
