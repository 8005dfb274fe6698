#!/bin/bash
# forward blend experiments: raster / binning tests, phase timing (timing build), kernel stats, bench
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-fwd}; mkdir -p $o
python -m pytest tests/test_gpu_raster.py tests/test_gpu_binning.py -x -q > $o/tests.log 2>&1; tail -2 $o/tests.log
ADGS_LIB=$R/ad-gs_amd/lib/libadgs_hip_timing.so python tools/fwd_phase_timing.py C3 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
python3 - $o/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("%-70s %5s %9.1f" % (r['Name'].replace('adgs::(anonymous namespace)::','').replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
cd $R
python bench.py --no-cpu-baseline --no-secondary > $o/bench.json 2> $o/bench.err; cut -c1-120 $o/bench.json
