#!/bin/bash
# Round-6 baseline lines: C3 headline (driver settings, no secondary), C5 / C4 iterations on one GPU.  gpurun --timeout 900 -- 'bash tools/gpu/r6_base.sh <tag>'
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-r6base}; mkdir -p $o; cd $R
timeout 400 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $o/bench_c3.json 2> $o/c3.err
timeout 400 python bench.py --steps 20 --warmup 5 --config C5 --no-secondary --no-cpu-baseline > $o/bench_c5.json 2> $o/c5.err
timeout 400 python bench.py --steps 20 --warmup 5 --config C4 --no-secondary --no-cpu-baseline > $o/bench_c4.json 2> $o/c4.err
for f in c3 c5 c4; do python - <<PY
import json
try:
    d = json.loads(open("$o/bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["unit"], d["ms_per_step"], d["config"].get("pipeline"), d.get("stages_ms"))
except Exception as e:
    print("$f failed", e)
PY
done
