#!/bin/bash
# Round-6 binning check: the binning / raster / graph / thread tests, then the C3 / C5 / C4 lines.  gpurun --timeout 1500 -- 'bash tools/gpu/r6_bin.sh <tag>'
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-r6bin}; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_binning.py tests/test_gpu_raster.py tests/test_gpu_graph_capacity.py tests/test_gpu_threads.py -m gpu -q -p no:cacheprovider -rf -x --durations=15 > $o/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|Error|error" $o/tests.log | cut -c1-300 | tail -30
grep -A18 "slowest" $o/tests.log | cut -c1-200
bash tools/gpu/r6_base.sh ${1:-r6bin}
if [ -n "$PARITY" ]; then timeout 600 python tools/parity_stats.py C2 C3 --out $o/parity_stats.json > $o/parity_stats.txt 2> $o/parity.err; grep "strict" $o/parity_stats.txt | cut -c1-330; fi
