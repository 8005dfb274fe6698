#!/bin/bash
# Profiles of `bench.py` (C3, driver settings): rocprofv3 kernel stats, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), SQ / GRBM
# counters of the blend kernels, the whole training iteration, the blend-kernel probe and wave timelines, the VALU microbenchmark.
# Counter passes never share a run with a trace domain; the program itself follows `--` (no env / shell hop).
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/${1:-prof}; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- $B > $o/stats.log 2>&1
export ADGS_BENCH_SKIP_STATS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -o fetch -- $B > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -o write -- $B > $o/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $o/sq -o sq -- $B > $o/sq.log 2>&1
# stall pass: LDS issue stalls and LDS activity, scalar activity, waves
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o/sq2 -o sq2 -- $B > $o/sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $o/sq3 -o sq3 -- $B > $o/sq3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $o/grbm -o grbm -- $B > $o/grbm.log 2>&1
rocprofv3 -L > $o/counters_available.txt 2>&1
unset ADGS_BENCH_SKIP_STATS
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ti -o ti -- python3 $R/examples/train_iteration.py --config C3 --iters 100 --json > $o/train_iteration.log 2>&1
# where the GPU idles inside an iteration: the same trace, previous kernel's end -> next kernel's start (tools/iteration_gaps.py)
python3 $R/tools/iteration_gaps.py $(find $o/ti -name "*kernel_trace.csv" | head -1) > $o/train_iteration_gaps.json 2> $o/gaps.err
cd $R
python tools/pmc_traffic.py $(find $o/fetch -name "*counter_collection.csv" | head -1) $(find $o/write -name "*counter_collection.csv" | head -1) $o/hbm_traffic_per_kernel.json $o/hbm_traffic_per_frame.json > $o/traffic.txt 2>&1
python tools/pmc_blend.py $o/pmc_blend_kernels.json $(find $o/sq $o/sq2 $o/sq3 $o/grbm -name "*counter_collection.csv") > $o/blend.txt 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
cp $(find $o/ti -name "*kernel_stats.csv" | head -1) $o/train_iteration_kernel_stats.csv
grep "^{" $o/stats.log | tail -1 > $o/bench_under_rocprof.json
grep "^{" $o/train_iteration.log | tail -1 > $o/train_iteration_under_rocprof.json
# the raw counter CSVs are tens of MB: keep only the summaries
rm -rf $o/fetch $o/write $o/sq $o/sq2 $o/sq3 $o/grbm $o/stats $o/ti
L=$R/ad-gs_amd/lib
ADGS_LIB=$L/libadgs_hip_probe.so python tools/blend_probe.py C3 20 > $o/blend_probe_c3.json 2> $o/probe.err
ADGS_LIB=$L/libadgs_hip_probe.so python tools/blend_probe.py C2 20 > $o/blend_probe_c2.json 2>> $o/probe.err
ADGS_LIB=$L/libadgs_hip_timeline.so python tools/wave_timeline.py C3 > $o/wave_timeline_c3.json 2> $o/timeline.err
ADGS_LIB=$L/libadgs_hip_timeline.so python tools/wave_timeline.py C2 > $o/wave_timeline_c2.json 2>> $o/timeline.err
ADGS_LIB=$L/libadgs_hip_timing.so python tools/blend_phase_timing.py C3 20 > $o/phase_timing_c3.json 2> $o/phase.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_hazards.hip -o /tmp/issue_hazards 2>/dev/null && /tmp/issue_hazards > $o/issue_hazards.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > $o/valu_rates.txt 2>&1
python tools/microbench/hbm_rates.py > $o/hbm_rates.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_latency.hip -o /tmp/gather_latency 2>/dev/null && /tmp/gather_latency > $o/gather_latency.txt 2>&1
python tools/stage_cache_experiment.py C3 > $o/stage_cache_experiment.json 2> $o/sce.err
tail -3 $o/stats.log | cut -c1-300; head -24 $o/traffic.txt; cat $o/blend.txt; head -14 $o/kernel_stats.csv | cut -c1-160
