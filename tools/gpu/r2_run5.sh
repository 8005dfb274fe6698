#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2e; mkdir -p $o
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_binning.py -x > $o/tests1.log 2>&1
echo "rc=$?" >> $o/tests1.log; tail -3 $o/tests1.log
timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_bucket.json 2> $o/bench_bucket.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config C5 > $o/bench_c5.json 2> $o/bench_c5.err
