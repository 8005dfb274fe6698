#!/bin/bash
# whole GPU suite + the driver-settings bench line: gpurun --timeout 3600 -- "bash tools/gpu/full_suite_and_bench.sh <tag>" -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-r3full}; mkdir -p $o; cd $R
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $o/gpu_tests.log 2>&1
tail -6 $o/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
try:
    d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], c["step_ms_hip_events"], "reruns", c.get("capacity_reruns"))
    print("stages", d.get("stages_ms"))
    print("roofline", d.get("roofline"))
    for r in d.get("other_configs", []):
        print(r.get("workload"), r.get("launch"), r.get("frames_per_s"))
    print("ref", (d.get("reference_api_path") or {}).get("frames_per_s"), "graph", (d.get("c3_graph_replay") or {}).get("frames_per_s"))
    ti = d.get("train_iteration"); print("train", ti if isinstance(ti, str) else (ti["ms_per_iteration"], ti["stage_ms"]))
    print("cpu", d.get("cpu_baseline"), d.get("parity"))
except Exception as e:
    print("bench failed", e)
PY
tail -3 $o/bench_driver.err
