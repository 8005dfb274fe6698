#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04train; mkdir -p $o; cd $R
for i in 1 2; do timeout 600 python examples/train_iteration.py --config C3 --iters 200 --json > $o/train_$i.json 2> $o/train_$i.err; done
python - <<PY
import json
for i in (1, 2):
    try:
        d = json.loads(open("$o/train_%d.json" % i).read().strip().splitlines()[-1]); print(d["ms_per_iteration"], d["stage_ms"], d["host_stage_ms"])
    except Exception as e:
        print("failed", e, open("$o/train_%d.err" % i).read()[-800:])
PY
ADGS_TEST_SEED_BASE=7000 ADGS_TEST_FACTORED_SEEDS=3 timeout 900 python -m pytest tests/test_gpu_exchange.py -q -p no:cacheprovider -k "factored_exchange_fuzz" > $o/fuzz2.log 2>&1; tail -5 $o/fuzz2.log
timeout 600 python -m pytest tests/test_gpu_env.py tests/test_gpu_loss.py tests/test_gpu_train_iteration.py -q -p no:cacheprovider -x > $o/tests.log 2>&1; tail -3 $o/tests.log
