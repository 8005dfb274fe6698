#!/bin/bash
# full GPU test suite, log to gpurun_out/<tag>/gpu_tests.log
cd $GRAFT_REPO_ROOT
tag=${1:-r2}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -p no:cacheprovider ${@:2} > gpurun_out/$tag/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/$tag/gpu_tests.log
tail -5 gpurun_out/$tag/gpu_tests.log
