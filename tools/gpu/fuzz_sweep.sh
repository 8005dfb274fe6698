#!/bin/bash
# The seeded fuzz tests of every component over ANOTHER seed range than the suite's default: gpurun --timeout 3000 -- 'bash tools/gpu/fuzz_sweep.sh <base> <n>'
R=$GRAFT_REPO_ROOT; b=${1:-1000}; n=${2:-60}; o=$R/gpurun_out/fuzz_$b; mkdir -p $o; cd $R
export ADGS_TEST_SEED_BASE=$b ADGS_TEST_SEEDS=$n ADGS_TEST_LARGE_SEEDS=$((n / 6)) ADGS_TEST_VARIANT_SEEDS=$n ADGS_TEST_ADVERSARIAL_SEEDS=$n ADGS_TEST_RENDER_SEEDS=$((n / 3)) \
  ADGS_TEST_DEFORM_SEEDS=$n ADGS_TEST_RAWSH_SEEDS=$n ADGS_TEST_ADAM_SEEDS=$n ADGS_TEST_LOSS_SEEDS=$n ADGS_TEST_ENV_SEEDS=$n ADGS_TEST_KNN_SEEDS=$((n * 3)) \
  ADGS_TEST_SIMPLE_KNN_SEEDS=$n ADGS_TEST_DENSIFY_SEEDS=$n ADGS_TEST_EXPAND_SEEDS=$n ADGS_TEST_FACTORED_SEEDS=$((n / 3))
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider -k "fuzz or random or seed" > $o/fuzz.log 2>&1
tail -5 $o/fuzz.log
