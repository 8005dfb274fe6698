#!/bin/bash
# new capacity / graph / regulariser tests, then the whole GPU suite
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3c; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_graph_capacity.py tests/test_gpu_loss.py -m gpu -q -p no:cacheprovider > $o/new_tests.log 2>&1
tail -15 $o/new_tests.log
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $o/gpu_tests.log 2>&1
tail -8 $o/gpu_tests.log
