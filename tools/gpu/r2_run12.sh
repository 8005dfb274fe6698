#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2l; mkdir -p $o
for m in 1 2 0; do
  ADGS_FWD_ORDER=$m timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $o/bench_order$m.json 2> $o/bench_order$m.err
done
ADGS_FWD_ORDER=2 timeout 600 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_raster.py -x > $o/tests.log 2>&1; tail -2 $o/tests.log
