#!/bin/bash
# profiles of `bench.py` (C3, driver settings) for profiles/r06: rocprofv3 kernel stats, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes),
# SQ counters of the blend kernels (wave states, LDS bank conflicts), the training iteration.  Counter passes never share a run with a trace
# domain; the program itself follows `--`.
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/${1:-r06_prof}; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
export ADGS_BENCH_PMC=0
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o stats -- $B > $o/stats.log 2>&1
export ADGS_BENCH_SKIP_STATS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -o fetch -- $B > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -o write -- $B > $o/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $o/sq -o sq -- $B > $o/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o/sq2 -o sq2 -- $B > $o/sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $o/sq3 -o sq3 -- $B > $o/sq3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $o/grbm -o grbm -- $B > $o/grbm.log 2>&1
unset ADGS_BENCH_SKIP_STATS
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ti -o ti -- python3 $R/examples/train_iteration.py --config C3 --iters 100 --json > $o/train_iteration.log 2>&1
python3 $R/tools/iteration_gaps.py $(find $o/ti -name "*kernel_trace.csv" | head -1) > $o/train_iteration_gaps.json 2> $o/gaps.err
cd $R
python tools/pmc_traffic.py $(find $o/fetch -name "*counter_collection.csv" | head -1) $(find $o/write -name "*counter_collection.csv" | head -1) $o/hbm_traffic_per_kernel.json $o/hbm_traffic_per_frame.json > $o/traffic.txt 2>&1
python tools/pmc_blend.py $o/pmc_blend_kernels.json $(find $o/sq $o/sq2 $o/sq3 $o/grbm -name "*counter_collection.csv") > $o/blend.txt 2>&1
# LDS bank conflicts of the two blend kernels (round 4: 23 % of the forward's LDS-active cycles)
python - <<PY > $o/lds_bank_conflicts.txt 2>&1
import csv, glob, collections
f = glob.glob("$o/sq3/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = "render_fwd_v2" if "render_fwd_v2_kernel" in r["Kernel_Name"] else "render_bwd_v2" if "render_bwd_v2_kernel" in r["Kernel_Name"] else None
    if k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: round(x) for c, x in v.items()}, "bank conflict cycles / LDS-active cycles = %.4f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
cp $(find $o/ti -name "*kernel_stats.csv" | head -1) $o/train_iteration_kernel_stats.csv
grep "^{" $o/stats.log | tail -1 > $o/bench_under_rocprof.json
grep "^{" $o/train_iteration.log | tail -1 > $o/train_iteration_under_rocprof.json
rm -rf $o/fetch $o/write $o/sq $o/sq2 $o/sq3 $o/grbm $o/stats $o/ti
python tools/microbench/hbm_rates.py > $o/hbm_rates.txt 2>&1
head -24 $o/traffic.txt; cat $o/blend.txt | head -30; cat $o/lds_bank_conflicts.txt; head -16 $o/kernel_stats.csv | cut -c1-150
