#!/bin/bash
# A/B of environment switches / library builds on one box: bash tools/gpu/ab_env.sh <tag> ; 200-step C3 eager runs, interleaved twice (edit the `run` lines)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/ab_env_$1; mkdir -p $o; cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-200} --warmup ${WARM:-30} --no-secondary --no-cpu-baseline $EXTRA > $o/$name.json 2> $o/$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/$name.json").read().strip().splitlines()[-1]); c = d["config"]; st = d.get("stages_ms") or {}
    print("%-22s value %7.1f median %.4f  fwd %.4f bwd %.4f pre %.4f" % ("$name", d["value"], c["step_ms_hip_events"]["median"], st.get("render_fwd", 0), st.get("render_bwd", 0), st.get("preprocess_fwd", 0)))
except Exception as e:
    print("$name failed", e)
PY
}
for rep in 1 2; do
run hint_$rep ADGS_X=0
run bottomup_$rep ADGS_FWD_ORDER=1
done
STEPS=20 WARM=5
for rep in 1 2 3; do run driver_$rep ADGS_X=0; done
