#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_b; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_loss.py tests/test_gpu_train_step.py -q -p no:cacheprovider -rf -s > $o/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $o/tests.log | cut -c1-300 | tail
grep -E "^it |param rel L2|rel L2" $o/tests.log | tail -60
timeout 600 python tools/host_profile.py C3 200 > $o/host_c3.txt 2>&1; head -45 $o/host_c3.txt
timeout 600 python tools/host_profile.py C1 300 > $o/host_c1.txt 2>&1; head -4 $o/host_c1.txt
