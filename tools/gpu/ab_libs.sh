#!/bin/bash
# A/B of whole-library builds on one box, every stage shown: `gpurun -- 'bash tools/gpu/ab_libs.sh default u8 default u8'` runs the C3 bench frame
# (16 cameras, 100 steps; CFG=C5 STEPS=20: another config) once per named build (ad-gs_amd/lib/libadgs_hip_<tag>.so; "default" = libadgs_hip.so), alternating.
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/ablibs; mkdir -p $o; cd $R
L=$R/ad-gs_amd/lib
i=0
for v in "$@"; do
  i=$((i+1))
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  ADGS_LIB=$lib ADGS_BENCH_PMC=0 python bench.py --steps ${STEPS:-100} --warmup 10 --no-secondary --no-cpu-baseline ${CFG:+--config $CFG} > $o/bench_${v}_$i.json 2> $o/bench_${v}_$i.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/bench_${v}_$i.json").read().strip().splitlines()[-1])
    print("$v", d["value"], " ".join("%s %.4f" % kv for kv in d["stages_ms"].items()))
except Exception as e:
    print("$v failed", e)
PY
done
