#!/bin/bash
# round 5, GPU call 6: the test the profile call stopped at (test_engine_tails_match_the_record_path), three times with its mode printed,
# then the whole -m gpu suite without -x.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c6
mkdir -p $O
export PYTHONFAULTHANDLER=1
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_round3_gpu.py -q -s -p no:cacheprovider -k "tails_match" > $O/tails_$i.txt 2>&1
  tail -3 $O/tails_$i.txt; grep -h "mode " $O/tails_$i.txt | head -3
done
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_full.txt 2>&1
tail -8 $O/pytest_full.txt
