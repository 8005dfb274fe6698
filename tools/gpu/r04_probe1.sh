#!/bin/bash
# round 4, first measurement: issue / hazard microbenchmark, per-phase wave timing of both blend kernels, baseline bench on this box
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-r04a}; mkdir -p $o; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_hazards.hip -o /tmp/issue_hazards 2> $o/ih_build.log && timeout 600 /tmp/issue_hazards > $o/issue_hazards.txt 2>&1
L=$R/ad-gs_amd/lib
ADGS_LIB=$L/libadgs_hip_timing.so timeout 600 python tools/blend_phase_timing.py C3 20 > $o/phase_timing_c3.json 2> $o/phase_c3.err
ADGS_LIB=$L/libadgs_hip_timing.so timeout 600 python tools/blend_phase_timing.py C2 20 > $o/phase_timing_c2.json 2> $o/phase_c2.err
timeout 900 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/bench_default.json 2> $o/bench_default.err
cat $o/issue_hazards.txt; cat $o/phase_timing_c3.json; tail -3 $o/phase_c3.err; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default", d["value"], "fwd", d["stages_ms"]["render_fwd"], "bwd", d["stages_ms"]["render_bwd"], d["config"]["step_ms_hip_events"]["median"])
PY
