#!/bin/bash
# A/B of library builds on one box: `gpurun -- 'bash tools/gpu/ab_variants.sh default lds0 default lds0'` runs the C3 bench frame
# (one camera, 100 steps) once per named build, alternating so that box-to-box and drift effects show, then the blend-kernel parity tests
# on the default build.  A build <tag> is ad-gs_amd/lib/libadgs_hip_<tag>.so (`make variant TAG=<tag> DEFS=...`); "default" = libadgs_hip.so.
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/ab; mkdir -p $o; cd $R
L=$R/ad-gs_amd/lib
for v in "$@"; do
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  ADGS_LIB=$lib python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", d["value"], "fwd", d["stages_ms"]["render_fwd"], "bwd", d["stages_ms"]["render_bwd"], d["config"]["step_ms_hip_events"]["median"])
except Exception as e:
    print("$v failed", e)
PY
done
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_gate_flips.py tests/test_gpu_random_configs.py -m gpu -x -q -p no:cacheprovider > $o/parity.log 2>&1
tail -4 $o/parity.log
