#!/bin/bash
# GPU suite (no -x: every failure is listed) + the driver-settings bench line: gpurun --timeout 2400 -- "bash tools/gpu/suite_and_bench.sh <tag> [pytest args]" -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; tag=${1:-r05}; shift; o=$R/gpurun_out/$tag; mkdir -p $o; cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -rf "$@" > $o/gpu_tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $o/gpu_tests.log | cut -c1-400 | tail -60
if [ -z "$NOBENCH" ]; then
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err
python - <<PY
import json
try:
    d = json.loads(open("$o/bench_driver.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], "roofline", d.get("roofline"))
    print("stages", d.get("stages_ms"))
    for r in d.get("other_configs", []):
        print(r.get("workload"), r.get("launch"), r.get("frames_per_s"))
    print("ref", (d.get("reference_api_path") or {}).get("frames_per_s"), "graph", (d.get("c3_graph_replay") or {}).get("frames_per_s"))
    print("sens", [(x.get("scene"), x.get("frames_per_s")) for x in d.get("scene_sensitivity", []) if isinstance(x, dict)])
    ti = d.get("train_iteration"); print("train", ti if not isinstance(ti, dict) else (ti["ms_per_iteration"], ti["stage_ms"], ti.get("host_stage_ms")))
    print("cpu", d.get("cpu_baseline"), d.get("parity"))
except Exception as e:
    print("bench failed", e)
PY
tail -3 $o/bench_driver.err
fi
