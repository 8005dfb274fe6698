#!/bin/bash
# the rasterizer fuzz families (tests/test_gpu_random_configs.py: strict gradient pass, no exemptions) over other seed ranges:
#   gpurun --timeout 3000 -- 'bash tools/gpu/fuzz_raster.sh <base> <n> [<base> <n> ...]'
R=$GRAFT_REPO_ROOT; cd $R
while [ $# -ge 2 ]; do
b=$1; n=$2; shift 2; o=$R/gpurun_out/r06_fuzz_$b; mkdir -p $o
ADGS_TEST_SEED_BASE=$b ADGS_TEST_SEEDS=$n ADGS_TEST_LARGE_SEEDS=$((n / 6)) ADGS_TEST_VARIANT_SEEDS=$n ADGS_TEST_ADVERSARIAL_SEEDS=$n \
  timeout 2700 python -m pytest tests/test_gpu_random_configs.py -q -p no:cacheprovider -rf > $o/fuzz.log 2>&1
echo "seed base $b, n $n:"; grep -E "^(FAILED|ERROR)|passed|failed" $o/fuzz.log | cut -c1-420 | tail -25
done
