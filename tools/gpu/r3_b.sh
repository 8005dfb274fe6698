#!/bin/bash
# Round 3, call 2: new capacity / graph tests first, then the whole GPU suite, then the bench with driver settings.
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3b; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_graph_capacity.py tests/test_gpu_loss.py -m gpu -x -q -p no:cacheprovider > $o/graph_tests.log 2>&1
tail -15 $o/graph_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
try:
    d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], c["step_ms_hip_events"], "reruns", c.get("capacity_reruns"), "settle", c.get("settle_steps"))
    print("idle", c.get("gpu_idle"))
    print("stages", d.get("stages_ms"))
    for k in ("other_configs",):
        for r in d.get(k, []):
            print(r.get("workload"), r.get("launch"), r.get("frames_per_s"), r.get("step_ms"), r.get("capacity_reruns"), r.get("failed"))
    print("ref path", d.get("reference_api_path"))
    print("c3 graph", d.get("c3_graph_replay"))
    print("roofline", d.get("roofline"))
except Exception as e:
    print("bench failed", e)
PY
tail -5 $o/bench_driver.err
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $o/gpu_tests.log 2>&1
tail -8 $o/gpu_tests.log
