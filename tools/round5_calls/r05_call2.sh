#!/bin/bash
# round 5, GPU call 2: the tests call 1 stopped in front of; what each launch chain costs the other (one chain dropped from the replayed
# plan: timing only); stream / graph-node priorities (audio chain low, video chain high) in graph replay and with eager launches.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c2
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_tattn_gpu.py tests/test_vconv_gpu.py -x -q -s -p no:cacheprovider > $O/pytest_a.txt 2>&1
tail -4 $O/pytest_a.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run default_1 X=1
run skip_audio MMD_SKIP_SID=1
run skip_video MMD_SKIP_SID=0
run prio_graph MMD_STREAM_PRIO=1
run prio_graph_nodeflag MMD_STREAM_PRIO=1 MMD_GRAPH_NODE_PRIO=1
run default_2 X=1
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown --no-graph"
run eager X=1
run eager_prio MMD_STREAM_PRIO=1
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
