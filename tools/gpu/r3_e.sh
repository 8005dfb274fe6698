#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3e; mkdir -p $o; cd $R
for ppl in 0 1 2 4; do for c in C2 C1; do
ADGS_V2_PPL=$ppl timeout 300 python bench.py --config $c --steps 200 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/b.json 2> $o/b.err
python - <<PY
import json
try:
    d = json.loads(open("$o/b.json").read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("ppl=$ppl $c", d["value"], d["ms_per_step"], "fwd", s.get("render_fwd"), "bwd", s.get("render_bwd"))
except Exception as e:
    print("ppl=$ppl $c failed", e)
PY
done; done
ADGS_V2_PPL=1 timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -p no:cacheprovider -k "small or long_tile or ragged or c2" > $o/ppl1_tests.log 2>&1; tail -3 $o/ppl1_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ti -o ti -- python3 $R/examples/train_iteration.py --config C3 --iters 100 --json > $o/ti.log 2>&1
cd $R; cp $(find $o/ti -name "*kernel_stats.csv" | head -1) $o/train_iteration_kernel_stats.csv; rm -rf $o/ti
head -40 $o/train_iteration_kernel_stats.csv | cut -c1-150
