#!/bin/bash
# kernel stats of bench.py with an experiment build of the library ($1 = tag of libadgs_hip_<tag>.so)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/probe_$1; mkdir -p $o
export ADGS_LIB=$R/ad-gs_amd/lib/libadgs_hip_$1.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
python3 - $o/kernel_stats.csv $1 <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(sys.argv[2], "%-60s %5s %9.1f" % (r['Name'].replace('adgs::(anonymous namespace)::','').replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
