#!/bin/bash
# Two ranks on ONE GPU over gloo (RCCL refuses duplicate devices): smoke test of the multi-rank paths of bench.py - batch-sharded sampling with
# the terminal all-gather timed apart, and the training step with the bucketed gradient all-reduce.  Throughput numbers are meaningless here.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/mr
mkdir -p $O
export MMD_DIST_BACKEND=gloo
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 2 > $O/sample2.log 2>&1
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --mode train --batch 2 --steps 2 --warmup 1 --no-graph > $O/train2.log 2>&1
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --mode train --batch 2 --steps 2 --warmup 1 > $O/train2_graph.log 2>&1
tail -2 $O/sample2.log | cut -c1-900; tail -2 $O/train2.log | cut -c1-600; tail -2 $O/train2_graph.log | cut -c1-600
