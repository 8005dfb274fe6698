#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_f; mkdir -p $o; cd $R
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_graph_capacity.py tests/test_gpu_deform.py tests/test_gpu_binning.py -q -p no:cacheprovider -rf > $o/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  +Assert" $o/tests.log | cut -c1-300 | tail -12
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $o/$name.json 2> $o/$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/$name.json").read().strip().splitlines()[-1]); c = d["config"]; st = d.get("stages_ms") or {}
    print("%-12s value %7.1f median %.4f  pre_fwd %.4f fwd %.4f bwd %.4f pre_bwd %.4f" % ("$name", d["value"], c["step_ms_hip_events"]["median"], st.get("preprocess_fwd", 0), st.get("render_fwd", 0), st.get("render_bwd", 0), st.get("preprocess_bwd", 0)))
except Exception as e:
    print("$name failed", e)
PY
}
for rep in 1 2 3; do
run new_$rep ADGS_X=1
run prev_$rep ADGS_LIB=$R/ad-gs_amd/lib/libadgs_hip_prev.so
done
