#!/bin/bash
# bench.py at a config under a list of environment settings: bash tools/gpu/r2_env_ab_cfg.sh tag C5 "A=1" "B=2 C=3" ...
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$1; mkdir -p $o; cfg=$2; shift; shift
i=0
for e in "" "$@"; do
  i=$((i+1))
  env $e python bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > $o/r$i.json 2> $o/r$i.err
  python - $o/r$i.json "$e" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms"]
print("%-44s %7.1f %s  %.3f ms/step  sort %.3f ranges %.3f dup %.3f fwd %.3f bwd %.3f  %s" % (sys.argv[2] or "(default)", d["value"], d["unit"], d["ms_per_step"], s["radix_sort"], s["tile_ranges"], s["duplicate_keys"], s["render_fwd"], s["render_bwd"], d["config"].get("pipeline","")[:48]))
PY
done
