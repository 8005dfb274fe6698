#!/bin/bash
# The round's other bench lines (profiles/rNN/bench_*.json): default settings, C4 / C5 iterations on one GPU, the N-GPU step as a one-rank
# RCCL dry run, parity statistics of C2 / C3 against the float64 oracle.  gpurun --timeout 2400 -- 'bash tools/gpu/bench_lines.sh <tag>'
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-lines}; mkdir -p $o; cd $R
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --config C4 --no-secondary > $o/bench_c4_1gpu.json 2> $o/c4.err
timeout 600 python bench.py --steps 20 --warmup 5 --config C5 --no-secondary > $o/bench_c5_1gpu.json 2> $o/c5.err
ADGS_BENCH_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_one_rank_rccl_dry_run.json 2> $o/rccl.err
timeout 900 python tools/parity_stats.py C2 C3 --out $o/parity_stats_default.json > $o/parity_stats_default.txt 2> $o/parity.err
ADGS_LIB=$R/ad-gs_amd/lib/libadgs_hip_precise.so timeout 900 python tools/parity_stats.py C2 C3 --out $o/parity_stats_precise_exp.json > $o/parity_stats_precise_exp.txt 2> $o/parity_precise.err
ADGS_BINNING=bucket timeout 600 python tools/tile_histogram.py C3 C5 C3:street C3:translucent C3:sky > $o/tile_histogram.txt 2> $o/hist.err; cp gpurun_out/tile_histogram.json $o/ 2>/dev/null
for f in default c4_1gpu c5_1gpu one_rank_rccl_dry_run; do python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["unit"], d["ms_per_step"], d["config"].get("capacity_reruns"))
except Exception as e:
    print("$f failed", e)
PY
done
tail -5 $o/parity_stats_default.txt | cut -c1-200
