#!/bin/bash
# round 5, GPU call 16: the -m gpu suite once more at the final build, exactly as the driver runs it (-x).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
export PYTHONFAULTHANDLER=1
timeout 640 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/c16/pytest_full.txt 2>&1
tail -4 gpurun_out/c16/pytest_full.txt
