#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2h; mkdir -p $o
L=$GRAFT_REPO_ROOT/ad-gs_amd/lib
for t in "" _pf1 _pf2; do
  ADGS_LIB=$L/libadgs_hip$t.so timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $o/bench$t.json 2> $o/bench$t.err
done
ADGS_LIB=$L/libadgs_hip_pf1.so timeout 900 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_raster.py tests/test_gpu_binning.py -x > $o/tests_pf1.log 2>&1
tail -2 $o/tests_pf1.log
