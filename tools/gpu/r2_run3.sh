#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2c; mkdir -p $o
python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_bench_multirank.py tests/test_gpu_gate_flips.py "tests/test_gpu_full_path.py::test_c4_three_camera_accumulation_factored_vs_conventional_and_oracle" tests/test_gpu_optim.py tests/test_gpu_env.py tests/test_gpu_exchange.py "tests/test_gpu_random_configs.py::test_v2_matches_classic_and_oracle_on_large_random_configs" -s > $o/tests.log 2>&1
echo "rc=$?" >> $o/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_20.json 2> $o/bench_20.err
python bench.py --gpus 1 --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $o/bench_300.json 2> $o/bench_300.err
for ct in 4 6 8; do
  ADGS_CELL_TILES=$ct python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_ct$ct.json 2> $o/bench_ct$ct.err
done
python bench.py --gpus 1 --steps 20 --warmup 5 --config C4 > $o/bench_c4.json 2> $o/bench_c4.err
python bench.py --gpus 1 --steps 20 --warmup 5 --config C5 > $o/bench_c5.json 2> $o/bench_c5.err
ADGS_BENCH_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_force.json 2> $o/bench_force.err
ADGS_BENCH_FORCE_COLLECTIVES=1 ADGS_DP_COLLECTIVE=rs_ag python bench.py --gpus 1 --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_force_rsag.json 2> $o/bench_force_rsag.err
tail -3 $o/tests.log
