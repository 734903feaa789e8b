#!/bin/bash
# Round 3, GPU call 38: the training bench line with its roofline object
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c38
timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > gpurun_out/c38/train.log 2>&1; tail -1 gpurun_out/c38/train.log | cut -c1-1800
