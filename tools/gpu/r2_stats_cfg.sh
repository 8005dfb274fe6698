#!/bin/bash
# rocprofv3 kernel stats of bench.py at another config: bash tools/gpu/r2_stats_cfg.sh C5 [extra bench args]
R=$GRAFT_REPO_ROOT; cfg=${1:-C5}; o=$R/gpurun_out/stats_$cfg; mkdir -p $o; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- python3 $R/bench.py --gpus 1 --steps 10 --warmup 4 --config $cfg --no-cpu-baseline --no-secondary "$@" > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
python3 - $o/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-72s %5s %9.1f us  %5.1f %%" % (r['Name'].replace('adgs::(anonymous namespace)::','').replace('void ','')[:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
tail -1 $o/stats.log | cut -c1-200
