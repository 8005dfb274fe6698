#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2j; mkdir -p $o
L=$GRAFT_REPO_ROOT/ad-gs_amd/lib
for t in "" _b5 _b6 _f6 _f8; do
  ADGS_LIB=$L/libadgs_hip$t.so timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $o/bench$t.json 2> $o/bench$t.err
done
echo ok
