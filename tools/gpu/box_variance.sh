#!/bin/bash
# headline variance probe: the driver's command (without the secondary lines) three times + a 200-step run, step series dumped
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_var_$1; mkdir -p $o; cd $R
lscpu | grep -E "Model name|^CPU\(s\)|MHz" | head -4
for i in 1 2 3; do
ADGS_BENCH_DUMP_STEPS=20 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $o/b$i.json 2> $o/b$i.err
python - <<PY
import json
d = json.loads(open("$o/b$i.json").read().strip().splitlines()[-1]); c = d["config"]
print("run $i value", d["value"], c["step_ms_hip_events"], "series", c.get("step_ms_series"))
PY
done
ADGS_BENCH_DUMP_STEPS=200 timeout 300 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $o/b200.json 2> $o/b200.err
python - <<PY
import json
d = json.loads(open("$o/b200.json").read().strip().splitlines()[-1]); c = d["config"]
s = c.get("step_ms_series")
print("200 steps value", d["value"], c["step_ms_hip_events"])
print("series", [round(x,2) for x in s[:60]])
PY
timeout 200 python tools/host_profile.py C3 200 2>&1 | head -3
