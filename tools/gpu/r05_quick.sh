#!/bin/bash
# quick validation: the rasterizer / thread / graph tests + the driver's bench command (all secondary lines)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_quick_$1; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_threads.py tests/test_gpu_graph_capacity.py tests/test_gpu_deform.py tests/test_gpu_train_step.py -q -p no:cacheprovider -x > $o/tests.log 2>&1
tail -3 $o/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "traffic", r.get("traffic"), r.get("traffic_measured_in_run"), r.get("traffic_source", "")[:200])
for x in d.get("other_configs", []): print(x.get("workload"), x.get("launch"), x.get("frames_per_s"))
print("knn", [(k.get("points"), k.get("ms"), k.get("roofline")) for k in d.get("knn_dist2", []) if "points" in k])
ti = d.get("train_iteration"); print("train", ti if not isinstance(ti, dict) else (ti["ms_per_iteration"], ti.get("host_stage_ms")))
print("cpu", d.get("cpu_baseline"))
PY
tail -3 $o/bench_driver.err
