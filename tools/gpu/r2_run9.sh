#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2i; mkdir -p $o
timeout 1200 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_deform.py tests/test_gpu_full_path.py tests/test_gpu_exchange.py tests/test_gpu_train_step.py tests/test_gpu_raster.py -x > $o/tests.log 2>&1
echo "rc=$?" >> $o/tests.log; tail -12 $o/tests.log
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $o/bench_rawscene.json 2> $o/bench_rawscene.err
ADGS_BENCH_RAW_SCENE=0 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $o/bench_norawscene.json 2> $o/bench_norawscene.err
tail -2 $o/*.err
