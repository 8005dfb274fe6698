#!/bin/bash
# A/B of library builds on the WHOLE training iteration and on the bench's secondary lines (one box): gpurun -- 'bash tools/gpu/ab_iteration.sh old default old default'
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/abiter; mkdir -p $o; cd $R
L=$R/ad-gs_amd/lib
i=0
for v in "$@"; do
  i=$((i+1))
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  ADGS_LIB=$lib python examples/train_iteration.py --iters 150 --json > $o/ti_${v}_$i.json 2> $o/ti_${v}_$i.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/ti_${v}_$i.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_iteration"], " ".join("%s %.3f" % kv for kv in d["stage_ms"].items()))
except Exception as e:
    print("$v failed", e)
PY
done
