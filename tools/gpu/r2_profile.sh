#!/bin/bash
# Round-2 profiles of `bench.py` (C3 defaults): rocprofv3 kernel stats, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), SQ / GRBM counters
# of the blend kernels.  Counter passes never share a run with a trace domain.  Output: gpurun_out/r2prof/ (copy the summaries to profiles/r02/).
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/r2prof; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- $B > $o/stats.log 2>&1
export ADGS_BENCH_SKIP_STATS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -o fetch -- $B > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -o write -- $B > $o/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $o/sq -o sq -- $B > $o/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $o/grbm -o grbm -- $B > $o/grbm.log 2>&1
cd $R
python tools/pmc_traffic.py $(find $o/fetch -name "*counter_collection.csv" | head -1) $(find $o/write -name "*counter_collection.csv" | head -1) $o/hbm_traffic_per_kernel.json $o/hbm_traffic_per_frame.json > $o/traffic.txt 2>&1
python tools/pmc_blend.py $(find $o/sq -name "*counter_collection.csv" | head -1) $(find $o/grbm -name "*counter_collection.csv" | head -1) $o/pmc_blend_kernels.json > $o/blend.txt 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
# the raw counter CSVs are tens of MB: keep only the summaries
rm -rf $o/fetch $o/write $o/sq $o/grbm $o/stats
tail -3 $o/stats.log; cat $o/traffic.txt | head -30; cat $o/blend.txt
