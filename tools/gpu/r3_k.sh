#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3k; mkdir -p $o; cd $R
timeout 600 python -m pytest tests/test_gpu_env.py tests/test_gpu_full_path.py -m gpu -x -q -p no:cacheprovider > $o/t.log 2>&1; tail -3 $o/t.log
python tools/env_bench.py > $o/env_bench.txt 2>&1; tail -6 $o/env_bench.txt
timeout 300 python examples/train_iteration.py --config C3 --iters 100 --json > $o/ti.json 2>/dev/null; python -c "
import json; d=json.load(open('$o/ti.json')); print(d['ms_per_iteration'], d['stage_ms'])"
