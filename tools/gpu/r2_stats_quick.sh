#!/bin/bash
# Quick A/B: the binning tests, then rocprofv3 kernel stats of bench.py at C3 (and at C5 with bucket binning forced).  Output: gpurun_out/$1/
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/${1:-quick}; mkdir -p $o
python -m pytest tests/test_gpu_binning.py -x -q > $o/tests.log 2>&1; tail -2 $o/tests.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- $B > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/stats
tail -1 $o/stats.log | cut -c1-300
head -24 $o/kernel_stats.csv | cut -d, -f1-4
cd $R
python bench.py --no-cpu-baseline --no-secondary > $o/bench.json 2> $o/bench.err; cut -c1-200 $o/bench.json
