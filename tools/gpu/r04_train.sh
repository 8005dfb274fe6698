#!/bin/bash
# the training iteration twice (device and host time per stage), with and without the fused image-loss node
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04train; mkdir -p $o; cd $R
timeout 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_train_step.py tests/test_gpu_env.py -q -p no:cacheprovider -x > $o/tests.log 2>&1; tail -3 $o/tests.log
for f in 0 1 0 1 0 1; do ADGS_FUSED_IMAGE_LOSSES=$f timeout 600 python examples/train_iteration.py --config C3 --iters 200 --json > $o/train_$f.json 2> $o/train_$f.err
python - <<PY
import json
try:
    d = json.loads(open("$o/train_$f.json").read().strip().splitlines()[-1]); print("fused $f", d["ms_per_iteration"], d["stage_ms"], d["host_stage_ms"], d["loss_first_last"])
except Exception as e:
    print("failed", e, open("$o/train_$f.err").read()[-800:])
PY
done
