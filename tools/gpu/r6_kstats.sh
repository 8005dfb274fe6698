#!/bin/bash
# Per-kernel times of a bench line under rocprofv3 (kernel trace only): gpurun -- 'bash tools/gpu/r6_kstats.sh <tag> [bench args]' -> gpurun_out/<tag>/kernel_stats.csv
R=$GRAFT_REPO_ROOT; tag=${1:-kst}; shift; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
grep "^{" $o/stats.log | tail -1 > $o/bench_under_rocprof.json
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$o/kernel_stats.csv")))
for r in rows[:26]:
    print("%-70s %6d avg %9.1f us  %5.1f %%" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
