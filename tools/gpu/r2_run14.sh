#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2m; mkdir -p $o
timeout 1200 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_full_path.py tests/test_gpu_deform.py tests/test_gpu_env.py tests/test_gpu_train_step.py tests/test_gpu_exchange.py -x -s > $o/tests.log 2>&1
echo "rc=$?" >> $o/tests.log; grep -E "passed|failed|^E  |rc=" $o/tests.log | head -20
timeout 300 python examples/train_iteration.py --config C3 --iters 20 > $o/train_iter.log 2>&1; tail -5 $o/train_iter.log
