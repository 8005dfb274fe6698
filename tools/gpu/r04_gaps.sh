#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04gaps; mkdir -p $o; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python3 $R/examples/train_iteration.py --config C3 --iters 60 --json > $o/train.json 2> $o/train.err
f=$(find $o/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/iteration_gaps.py $f > $o/gaps.json 2> $o/gaps.err; head -c 6000 $o/gaps.json; rm -rf $o/trace
cd $R
ADGS_TEST_SEED_BASE=7000 ADGS_TEST_FACTORED_SEEDS=3 ADGS_TEST_ADVERSARIAL_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_random_configs.py -q -p no:cacheprovider -k "factored_exchange_fuzz or adversarial" > $o/fuzz2.log 2>&1; tail -5 $o/fuzz2.log
