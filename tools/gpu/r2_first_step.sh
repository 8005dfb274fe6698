#!/bin/bash
# Driver settings three times + default once: is the first timed step still the slow one?
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-fs}; mkdir -p $o
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $o/d$i.json 2> $o/d$i.err; python - $o/d$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["step_ms_hip_events"])
PY
done
python bench.py --no-cpu-baseline --no-secondary > $o/long.json 2> $o/long.err; python - $o/long.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["step_ms_hip_events"])
PY
