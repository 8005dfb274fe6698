#!/bin/bash
# bench the C3 frame under environment settings: bash tools/gpu/r04_env.sh "NAME=VALUE ..." "NAME=VALUE ..." (use "-" for no setting)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${OUT:-r04env}; mkdir -p $o; cd $R
i=0
for e in "$@"; do
  i=$((i+1)); [ "$e" = "-" ] && e=""
  env $e python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/bench_$i.json 2> $o/bench_$i.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$i.json").read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("%-28s %7.1f fwd %.4f bwd %.4f bin %.4f step %.4f" % ("$e" or "default", d["value"], s["render_fwd"], s["render_bwd"], s["scan"] + s["duplicate_keys"] + s["radix_sort"] + s["tile_ranges"], d["config"]["step_ms_hip_events"]["median"]))
except Exception as ex:
    print("$e failed", ex)
PY
done
