#!/bin/bash
# round 4 A/B: parity of the blend kernels on the default build first (fast fail), then the C3 bench frame per named build, alternating
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${OUT:-r04ab}; mkdir -p $o; cd $R
L=$R/ad-gs_amd/lib
if [ "${PARITY:-1}" = 1 ]; then
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_gate_flips.py tests/test_gpu_binning.py -m gpu -x -q -p no:cacheprovider > $o/parity.log 2>&1
tail -5 $o/parity.log
fi
for v in "$@"; do
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  ADGS_LIB=$lib timeout 600 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("%-10s %7.1f fwd %.4f bwd %.4f pre %.4f pbwd %.4f dfwd %.4f dbwd %.4f bin %.4f step median %.4f" % ("$v", d["value"], s["render_fwd"], s["render_bwd"], s.get("preprocess_fwd", 0), s.get("preprocess_bwd", 0),
          s.get("deform_fwd", 0), s.get("deform_bwd", 0), s.get("scan", 0) + s.get("duplicate_keys", 0) + s.get("radix_sort", 0) + s.get("tile_ranges", 0), d["config"]["step_ms_hip_events"]["median"]))
except Exception as e:
    print("$v failed", e)
PY
done
if [ -f $L/libadgs_hip_timing.so ] && [ "${TIMING:-1}" = 1 ]; then
ADGS_LIB=$L/libadgs_hip_timing.so timeout 600 python tools/blend_phase_timing.py C3 20 > $o/phase_timing_c3.json 2> $o/phase_c3.err
python - <<PY
import json
d = json.load(open("$o/phase_timing_c3.json"))
f, b = d["render_fwd_v2"], d["render_bwd_v2"]
print("fwd cycles/wave", f["cycles_per_wave"], {k: f[k]["share"] for k in ("key_stream_scan", "filter_record_test", "splat_gather_staging", "blend_loop")}, "wait/scan round", f["cycles_waiting_per_scan_super_round"], "wait/filter round", f["cycles_waiting_per_filter_round"])
print("bwd cycles/wave", b["cycles_per_wave"], {k: b[k]["share"] for k in ("chunk_header_wait", "id_and_splat_gather", "entry_loop")}, "entry", b["entry_loop"]["cycles_per_entry"], "reduce", b["reduction_and_atomic"]["cycles_per_entry"])
PY
fi
