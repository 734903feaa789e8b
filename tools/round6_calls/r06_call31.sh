#!/bin/bash
# round 6, GPU call 31: HIP runtime switches that touch how a replayed graph's launches reach the hardware queues (timing only): the step
# and the queue-overlap probe under each
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c31
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
probe() { name=$1; shift; env "$@" timeout 200 python tools/queue_overlap_probe.py 64 > $O/probe_$name.txt 2>&1; }
run base_1 X=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq1 GPU_MAX_HW_QUEUES=1
run gq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run gq8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run devkarg0 HIP_FORCE_DEV_KERNARG=0
run devkarg1 HIP_FORCE_DEV_KERNARG=1
run optflush0 AMD_OPT_FLUSH=0
run optflush1 AMD_OPT_FLUSH=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run dynq1 DEBUG_HIP_DYNAMIC_QUEUES=1
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run gbatch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run gbatch1k DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run base_2 X=1
probe base X=1
probe hwq8 GPU_MAX_HW_QUEUES=8
probe pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
probe optflush0 AMD_OPT_FLUSH=0
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(p))
        print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  finite {d['config'].get('finite')}")
    except Exception as e:
        print(os.path.basename(p), "unreadable", str(e)[:80])
PY
cat $O/ab_lines.txt; tail -n 7 $O/probe_*.txt
