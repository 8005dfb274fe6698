#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3g; mkdir -p $o; cd $R
timeout 1200 python -m pytest tests/test_gpu_knn_points.py tests/test_gpu_optim.py tests/test_gpu_train_step.py tests/test_gpu_env.py -m gpu -q -x -p no:cacheprovider > $o/tests.log 2>&1
tail -6 $o/tests.log
timeout 300 python examples/train_iteration.py --config C3 --iters 210 --json > $o/train_iteration.json 2> $o/train_iteration.err; tail -2 $o/train_iteration.err; cat $o/train_iteration.json
timeout 300 python tools/densify_bench.py > $o/densify_bench.txt 2>&1; tail -8 $o/densify_bench.txt
