#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2final; mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_settings.json 2> $o/bench_driver_settings.err
python bench.py > $o/bench_default.json 2> $o/bench_default.err
tail -2 $o/bench_driver_settings.err
