#!/bin/bash
# round 2, run 1: new full-path tests (stats printed), parity statistics vs f64, baseline bench at driver settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_full_path.py -q -s -m gpu -p no:cacheprovider > gpurun_out/r2/full_path.log 2>&1
echo "full_path rc=$?" >> gpurun_out/r2/full_path.log
python tools/parity_stats.py C2 C3 --out gpurun_out/r2/parity_stats_default.json > gpurun_out/r2/parity_stats_default.log 2>&1
ADGS_LIB=$GRAFT_REPO_ROOT/ad-gs_amd/lib/libadgs_hip_precise.so python tools/parity_stats.py C2 C3 --out gpurun_out/r2/parity_stats_precise.json > gpurun_out/r2/parity_stats_precise.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/bench_20.json 2> gpurun_out/r2/bench_20.err
python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r2/bench_300.json 2> gpurun_out/r2/bench_300.err
ADGS_LIB=$GRAFT_REPO_ROOT/ad-gs_amd/lib/libadgs_hip_precise.so python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r2/bench_300_precise.json 2> gpurun_out/r2/bench_300_precise.err
echo done
