#!/bin/bash
# the driver's bench line (all secondary keys) on the current default build, after a quick A/B of named variant builds
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${OUT:-r04full}; mkdir -p $o; cd $R
if [ $# -gt 0 ]; then PARITY=0 TIMING=0 OUT=${OUT:-r04full} bash tools/gpu/r04_ab.sh "$@"; fi
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
print("driver line:", d["value"], d["ms_per_step"], "roofline", d.get("roofline"))
for k in ("other_configs", "reference_api_path", "c3_graph_replay", "scene_sensitivity", "knn_dist2"):
    v = d.get(k)
    if isinstance(v, list):
        for x in v: print(k, {kk: x[kk] for kk in list(x)[:12]} if isinstance(x, dict) else x)
    elif isinstance(v, dict): print(k, {kk: v[kk] for kk in list(v)[:10]})
    else: print(k, v)
ti = d.get("train_iteration")
print("train_iteration", ti if not isinstance(ti, dict) else {k: ti[k] for k in list(ti)[:14]})
print("parity", d.get("parity")); print("cpu_baseline", d.get("cpu_baseline"))
PY
