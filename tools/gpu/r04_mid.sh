#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${OUT:-r04mid}; mkdir -p $o; cd $R
timeout 1200 python -m pytest tests/test_gpu_loss.py tests/test_gpu_env.py tests/test_gpu_optim.py tests/test_gpu_train_step.py tests/test_gpu_graph_capacity.py -m gpu -x -q -p no:cacheprovider > $o/tests.log 2>&1; tail -4 $o/tests.log
L=$R/ad-gs_amd/lib
ADGS_LIB=$L/libadgs_hip_timing.so timeout 600 python tools/blend_phase_timing.py C3 20 > $o/phase_timing_c3.json 2> $o/phase_c3.err
python - <<PY
import json
d = json.load(open("$o/phase_timing_c3.json"))
f, b = d["render_fwd_v2"], d["render_bwd_v2"]
print("fwd", json.dumps(f))
print("bwd", json.dumps(b))
PY
OUT=${OUT:-r04mid} bash tools/gpu/r04_full.sh
