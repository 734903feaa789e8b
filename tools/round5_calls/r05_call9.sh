#!/bin/bash
# round 5, GPU call 9: the batch-row failure of test_engine_tails_match_the_record_path appeared only inside long pytest processes (1 of 3) and never
# alone (0 of 16): repeat its file (and the files in front of it in suite order) in one process several times, with the mode in the message.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c9
mkdir -p $O
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6; do
  timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_round3_gpu.py -q -p no:cacheprovider -k "not attention and not attn" > $O/run_$i.txt 2>&1
  tail -2 $O/run_$i.txt; grep -h "mode " $O/run_$i.txt | head -2
done
