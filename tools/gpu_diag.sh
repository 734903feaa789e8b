#!/bin/bash
# Round-2 crash diagnosis: the committed round-1 HEAD (copied to _head_copy/) under faulthandler, with and without the cyclic GC,
# then every GPU test file in its own process.  usage: gpurun -- 'bash tools/gpu_diag.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/diag
mkdir -p $O
export PYTHONFAULTHANDLER=1
cd _head_copy
timeout 900 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/head_full.log 2>&1
echo "rc=$?" >> $O/head_full.log
MMD_DIAG_NOGC=1 PYTHONPATH=$GRAFT_REPO_ROOT/tools/diag_nogc timeout 900 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/head_nogc.log 2>&1
echo "rc=$?" >> $O/head_nogc.log
for f in test_model_gpu test_ops_gpu test_sampling_api_gpu test_sr_gpu test_train_gpu test_trainloop_gpu; do
  timeout 900 python -X faulthandler -m pytest tests/$f.py -q -m gpu -p no:cacheprovider > $O/head_$f.log 2>&1
  echo "rc=$?" >> $O/head_$f.log
done
cd ..
tail -5 $O/*.log
