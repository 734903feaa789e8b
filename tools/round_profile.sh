set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/f/pytest.txt
timeout 600 python bench.py --breakdown-out gpurun_out/f/breakdown.json > gpurun_out/f/bench_default.log 2>&1
timeout 400 python bench.py --mode dpm --batch 2 --steps 2 --warmup 1 > gpurun_out/f/bench_dpm.log 2>&1
timeout 400 python bench.py --mode train --batch 8 --steps 3 --warmup 1 > gpurun_out/f/bench_train.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/f/kt -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > gpurun_out/f/kt.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/f/pmc_fetch -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > gpurun_out/f/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/f/pmc_write -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > gpurun_out/f/pmc_write.log 2>&1
ls -la gpurun_out/f gpurun_out/f/kt | head -30
tail -3 gpurun_out/f/pytest.txt
