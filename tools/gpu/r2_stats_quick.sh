#!/bin/bash
# Quick A/B: selected tests ($2, default the binning tests), then rocprofv3 kernel stats of bench.py at C3 and a plain bench run.  Output: gpurun_out/$1/
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/${1:-quick}; mkdir -p $o
python -m pytest ${2:-tests/test_gpu_binning.py} -x -q > $o/tests.log 2>&1; tail -2 $o/tests.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- $B > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
python3 - $o/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-70s %5s %9.1f" % (r['Name'].replace('adgs::(anonymous namespace)::','').replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
cd $R
python bench.py --no-cpu-baseline --no-secondary > $o/bench.json 2> $o/bench.err; cut -c1-200 $o/bench.json
