#!/bin/bash
# Round 3, call 1: sustained VALU clock, blend-kernel variants A/B, parity of the new default, pair statistics.
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3a; mkdir -p $o; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > $o/valu_rates.txt 2>&1
L=$R/ad-gs_amd/lib
for v in r2 old3 default nofwdexec nobwdinit; do
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  ADGS_LIB=$lib python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", d["value"], d["stages_ms"], d["config"]["step_ms_hip_events"])
except Exception as e:
    print("$v failed", e)
PY
done
ADGS_LIB=$L/libadgs_hip_probe.so python tools/blend_probe.py C3 20 > $o/probe_c3.json 2> $o/probe_c3.err
ADGS_LIB=$L/libadgs_hip_probe.so python tools/blend_probe.py C2 20 > $o/probe_c2.json 2> $o/probe_c2.err
cat $o/probe_c3.json
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_gate_flips.py tests/test_gpu_binning.py -m gpu -x -q -p no:cacheprovider > $o/parity.log 2>&1
tail -5 $o/parity.log
head -12 $o/valu_rates.txt
