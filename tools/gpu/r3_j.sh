#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3j; mkdir -p $o; cd $R
for i in 1 2; do
python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/b.json 2> $o/b.err
python - <<PY
import json
d = json.loads(open("$o/b.json").read().strip().splitlines()[-1])
print(d["value"], d["stages_ms"])
PY
done
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_deform.py -m gpu -x -q -p no:cacheprovider > $o/t.log 2>&1; tail -3 $o/t.log
