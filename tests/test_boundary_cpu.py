"""Drop-in boundary on the CPU: the import surface of the two target scripts resolves against this package, the
N > 1 benchmark entry brings its own ranks up, and the evaluator counterpart fails loudly only when called.

The names below are restated as data (what `py_scripts/multimodal_train.py:7-17` and `py_scripts/multimodal_sample_sr.py:13-27` import
and the `dist_util.` / `logger.` attributes their bodies touch); the scripts' text is not kept here."""
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRAIN_SCRIPT_IMPORTS = {
    "mm_diffusion": ["dist_util", "logger"],
    "mm_diffusion.multimodal_datasets": ["load_data"],
    "mm_diffusion.resample": ["create_named_schedule_sampler"],
    "mm_diffusion.multimodal_script_util": ["model_and_diffusion_defaults", "create_model_and_diffusion", "args_to_dict", "add_dict_to_argparser"],
    "mm_diffusion.multimodal_train_util": ["TrainLoop"],
    "mm_diffusion.common": ["set_seed_logger_random"],
}
SAMPLE_SCRIPT_IMPORTS = {
    "mm_diffusion": ["dist_util", "logger"],
    "mm_diffusion.multimodal_script_util": ["model_and_diffusion_defaults", "create_model_and_diffusion", "add_dict_to_argparser", "args_to_dict"],
    "mm_diffusion.script_util": ["image_sr_model_and_diffusion_defaults", "image_sr_create_model_and_diffusion"],
    "mm_diffusion.common": ["set_seed_logger_random", "save_audio", "save_img", "save_multimodal", "delete_pkl"],
    "mm_diffusion.multimodal_dpm_solver_plus": ["DPM_Solver"],
    "mm_diffusion.dpm_solver_plus": ["DPM_Solver"],
    "mm_diffusion.evaluator": ["eval_multimodal"],
}
MODULE_ATTRS_USED = {          # attribute accesses in the script bodies
    "mm_diffusion.dist_util": ["setup_dist", "dev", "load_state_dict"],
    "mm_diffusion.logger": ["configure", "log"],
}


@pytest.mark.parametrize("surface", [TRAIN_SCRIPT_IMPORTS, SAMPLE_SCRIPT_IMPORTS, MODULE_ATTRS_USED], ids=["train", "sample_sr", "attrs"])
def test_script_import_surface_resolves(surface):
    for mod, names in surface.items():
        m = importlib.import_module(mod)
        for n in names:
            obj = getattr(m, n, None)
            if obj is None:                       # `from mm_diffusion import dist_util` style: a submodule
                obj = importlib.import_module(f"{mod}.{n}")
            assert obj is not None, f"{mod}.{n}"


def test_evaluator_counterpart_does_not_cost_the_samples(monkeypatch):
    """The reference sampling script calls eval_multimodal AFTER the sampling loop (sample_sr.py:268): there the counterpart logs an error,
    warns and returns a sentinel (never an empty dict) instead of throwing the finished samples away; a metrics-only run (eval.py) and
    MMD_EVAL_STRICT=1 make it an error; the flag parser gives the early notice once per path."""
    import argparse
    from mm_diffusion import evaluator, multimodal_script_util as msu
    monkeypatch.delenv("MMD_EVAL_STRICT", raising=False)
    monkeypatch.setattr(sys, "argv", ["py_scripts/multimodal_sample_sr.py"])
    with pytest.warns(RuntimeWarning, match="out of scope"):
        assert evaluator.eval_multimodal("/ref", "/fake", eval_num=8) == {"unavailable": True}
    monkeypatch.setattr(sys, "argv", ["py_scripts/eval.py"])            # nothing but the metrics at stake: fail
    with pytest.raises(evaluator.EvaluatorUnavailable, match="out of scope"):
        evaluator.eval_multimodal("/ref", "/fake", eval_num=8)
    monkeypatch.setenv("MMD_EVAL_STRICT", "0")                           # explicit override
    with pytest.warns(RuntimeWarning):
        assert evaluator.eval_multimodal("/ref", "/fake")["unavailable"] is True
    monkeypatch.setattr(sys, "argv", ["py_scripts/multimodal_sample_sr.py"])
    monkeypatch.setenv("MMD_EVAL_STRICT", "1")
    with pytest.raises(evaluator.EvaluatorUnavailable, match="out of scope"):
        evaluator.eval_multimodal("/ref", "/fake", eval_num=8)
    import inspect
    sig = inspect.signature(evaluator.eval_multimodal)
    assert list(sig.parameters) == ["real_path", "fake_path", "video_size", "eval_num"] and sig.parameters["eval_num"].default == 2048
    evaluator._warned.clear()
    ns = argparse.Namespace(ref_path="/data/landscape/test", num_channels=128)
    with pytest.warns(RuntimeWarning, match="out of scope"):
        assert msu.args_to_dict(ns, ["num_channels"]) == {"num_channels": 128}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                  # the notice is given once per path
        assert msu.args_to_dict(ns, ["num_channels"]) == {"num_channels": 128}
        assert msu.args_to_dict(argparse.Namespace(num_channels=64), ["num_channels"]) == {"num_channels": 64}


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher environment: bench.py spawns the two ranks itself (torch.distributed.run on a free
    127.0.0.1 port), the ranks form a process group (gloo here: no GPU) and rank 0 prints one JSON line that records what the
    collective backend saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"


def test_bench_refuses_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
