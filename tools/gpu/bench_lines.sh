#!/bin/bash
# The round's bench lines (profiles/rNN/bench_*.json): driver settings with every secondary line, default settings, C4 / C5 iterations on one GPU (one and
# two camera streams), the N-GPU step as a one-rank RCCL dry run, parity statistics of C2 / C3 (strict gradient pass).  gpurun --timeout 2700 -- 'bash tools/gpu/bench_lines.sh <tag>'
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-lines}; mkdir -p $o; cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_driver_settings.json 2> $o/bench_driver.err
timeout 900 python bench.py --no-secondary > $o/bench_default.json 2> $o/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --config C4 --no-secondary > $o/bench_c4_1gpu.json 2> $o/c4.err
timeout 600 python bench.py --steps 20 --warmup 5 --config C5 --no-secondary > $o/bench_c5_1gpu.json 2> $o/c5.err
ADGS_BENCH_STREAMS=2 timeout 600 python bench.py --steps 20 --warmup 5 --config C4 --no-secondary --no-cpu-baseline > $o/bench_c4_1gpu_2streams.json 2> $o/c4s.err
ADGS_BENCH_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_one_rank_rccl_dry_run.json 2> $o/rccl.err
timeout 900 python tools/parity_stats.py C2 C3 --out $o/parity_stats_strict.json > $o/parity_stats_strict.txt 2> $o/parity.err
for f in driver_settings default c4_1gpu c5_1gpu c4_1gpu_2streams one_rank_rccl_dry_run; do python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["unit"], d["ms_per_step"], d["config"].get("capacity_reruns"))
except Exception as e:
    print("$f failed", e)
PY
done
tail -5 $o/parity_stats_strict.txt | cut -c1-200
