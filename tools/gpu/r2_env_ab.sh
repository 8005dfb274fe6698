#!/bin/bash
# bench.py (200 steps) under a list of environment settings: bash tools/gpu/r2_env_ab.sh tag "A=1" "B=2 C=3" ...
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$1; mkdir -p $o; shift
i=0
for e in "" "$@"; do
  i=$((i+1))
  env $e python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $o/r$i.json 2> $o/r$i.err
  python - $o/r$i.json "$e" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms"]
print("%-40s %7.1f fps  median %.4f ms  fwd %.3f bwd %.3f" % (sys.argv[2] or "(default)", d["value"], d["config"]["step_ms_hip_events"]["median"], s["render_fwd"], s["render_bwd"]))
PY
done
