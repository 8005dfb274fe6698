#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3d; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_graph_capacity.py tests/test_gpu_train_step.py -m gpu -q -p no:cacheprovider > $o/new_tests.log 2>&1
tail -6 $o/new_tests.log
timeout 300 python examples/train_iteration.py --config C3 --iters 210 --json > $o/train_iteration.json 2> $o/train_iteration.err; tail -2 $o/train_iteration.err; cat $o/train_iteration.json
for c in C2 C1; do
timeout 300 python bench.py --config $c --steps 200 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_$c.json 2> $o/bench_$c.err
python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$c.json").read().strip().splitlines()[-1])
    print("$c", d["value"], d["ms_per_step"], d["stages_ms"], d["config"].get("gpu_idle"))
except Exception as e:
    print("$c failed", e)
PY
done
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
try:
    d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], c["step_ms_hip_events"], "reruns", c.get("capacity_reruns"), "settle", c.get("settle_steps"))
    print("train_iteration", d.get("train_iteration"))
    print("sens", [(r.get("workload"), r.get("frames_per_s")) for r in d.get("scene_sensitivity", [])])
except Exception as e:
    print("bench failed", e)
PY
tail -3 $o/bench_driver.err
