#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of any python script: gpurun -- 'bash tools/gpu/r6_kstats_cmd.sh <tag> tools/inference_bench.py C3'
R=$GRAFT_REPO_ROOT; tag=${1:-kst}; shift; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
script=$R/$1; shift
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- python3 $script "$@" > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$o/kernel_stats.csv")))
for r in rows[:${TOP:-30}]:
    print("%-64s %6d avg %9.1f us  %5.1f %%" % (r["Name"][:64].replace("adgs::(anonymous namespace)::","").replace("void ",""), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
