"""The N > 1 legs of bench.py on ONE GPU (`-m gpu`): two ranks over gloo sharing cuda:0 (RCCL refuses duplicate devices; the driver's
8-GPU run uses RCCL through the same code).  What is checked is the multi-rank plumbing the scaling run depends on, not throughput:
bench.py launches its own ranks, the collective spans both, the JSON line carries n_gpus / ranks_seen / terminal_all_gather_ms, and
every rank's trajectory is the one a lone process with that rank's seeds produces (batch-sharded sampling has no in-loop collective:
/root/reference/py_scripts/multimodal_sample.py:94-131 shards by rank and gathers once at the end)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_sampling_json_and_per_rank_trajectories(tmp_path):
    """`python bench.py --gpus 2` with no launcher: both ranks run configs[1]'s step at batch 2, rank 0 prints ONE line with the
    whole-job rate; rank r's final sample == the lone-process run seeded like rank r, bitwise."""
    d2 = str(tmp_path / "two")
    res = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--no-breakdown", "--dump-final", d2)
    cfg = res["config"]
    assert res["n_gpus"] == 2 and cfg["ranks_seen"] == 2 and cfg["backend"] == "gloo" and cfg["global_batch"] == 4
    assert cfg["finite"] is True and cfg["terminal_all_gather_ms"] > 0 and res["value"] > 0 and res["scaling"] == "weak"
    assert abs(res["value"] - 4 * 1000.0 / res["ms_per_step"]) < 1e-6 * res["value"]       # whole-job pairs per max-over-ranks step time
    two = [torch.load(os.path.join(d2, f"rank{r}.pt")) for r in (0, 1)]
    assert not torch.equal(two[0]["video"], two[1]["video"])                                 # the ranks do not sample the same noise
    for r in (0, 1):
        d1 = str(tmp_path / f"one{r}")
        one = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2", "--no-breakdown", "--no-cpu-baseline", "--as-rank", str(r),
                     "--dump-final", d1)
        assert one["n_gpus"] == 1 and one["config"]["terminal_all_gather_ms"] is None
        lone = torch.load(os.path.join(d1, f"rank{r}.pt"))
        assert torch.equal(lone["video"], two[r]["video"]) and torch.equal(lone["audio"], two[r]["audio"]), f"rank {r}"


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_training_step_with_bucketed_all_reduce(graph):
    """configs[3] with two ranks: the eager step (gradient buckets all-reduced on the side stream during the backward) and the
    graph-captured step; the line carries n_gpus 2 and a finite loss."""
    argv = ["--gpus", "2", "--mode", "train", "--batch", "2", "--steps", "2", "--warmup", "1"] + ([] if graph else ["--no-graph"])
    res = _bench(*argv)
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["config"]["global_batch"] == 4
    assert math.isfinite(res["config"]["loss"]) and res["value"] > 0
