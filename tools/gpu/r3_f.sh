#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3f; mkdir -p $o; cd $R
timeout 1200 python -m pytest tests/test_gpu_binning.py tests/test_gpu_raster.py tests/test_gpu_bench_multirank.py -m gpu -q -x -p no:cacheprovider > $o/tests.log 2>&1
tail -6 $o/tests.log
for c in C3 C5; do for b in bucket sort; do
ADGS_BINNING=$b timeout 600 python bench.py --config $c --steps 40 --warmup 6 --no-secondary --no-cpu-baseline > $o/b_${c}_$b.json 2> $o/b.err
python - <<PY
import json
try:
    d = json.loads(open("$o/b_${c}_$b.json").read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("$c $b", d["value"], d["ms_per_step"], {k: s[k] for k in ("scan", "duplicate_keys", "radix_sort", "tile_ranges", "render_fwd") if k in s}, d["config"].get("pipeline", "")[:40])
except Exception as e:
    print("$c $b failed", e); print(open("$o/b.err").read()[-600:])
PY
done; done
