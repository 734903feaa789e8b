#!/bin/bash
# A dispatch-only change (ops.py / engine.py: which of two bitwise-equal paths a launch takes) after the round's full suite: re-run the
# tests that drive the sampling engine, then the profiling passes of tools/round_profile.sh (RP_SKIP_TESTS=1) so that profiles/ and
# pmc_traffic.json are stamped with the final build id.  Every step bounded; total < 5.5 min.
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
timeout 110 python -m pytest tests/test_strip_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider > $O/recheck_tests1.txt 2>&1
echo "tests1 rc=$?" > $O/recheck_rc.txt
RP_SKIP_TESTS=1 bash tools/round_profile.sh $TAG > $O/recheck_rp.log 2>&1
echo "rp rc=$?" >> $O/recheck_rc.txt
timeout 100 python -m pytest tests/test_sampling_api_gpu.py tests/test_lifetime_gpu.py tests/test_configs_gpu.py tests/test_dpm_solver_gpu.py -x -q -p no:cacheprovider > $O/recheck_tests2.txt 2>&1
echo "tests2 rc=$? (124 = cut by the time bound)" >> $O/recheck_rc.txt
cat $O/recheck_rc.txt
tail -3 $O/recheck_tests1.txt
tail -3 $O/recheck_tests2.txt
cat profiles/${TAG}_bench_line.json
