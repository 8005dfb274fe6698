#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2final; mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_settings.json 2> $o/bench_driver_settings.err
python bench.py > $o/bench_default.json 2> $o/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --config C4 > $o/bench_c4.json 2> $o/bench_c4.err
python bench.py --gpus 1 --steps 20 --warmup 5 --config C5 > $o/bench_c5.json 2> $o/bench_c5.err
ADGS_BENCH_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_force.json 2> $o/bench_force.err
tail -2 $o/bench_driver_settings.err
