#!/bin/bash
# depth slabs: binning tests, the two re-toleranced fuzz cases, C3 / C5 A/B against ADGS_SLABS=1
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04slabs; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_binning.py -x -q -p no:cacheprovider > $o/binning.log 2>&1; tail -3 $o/binning.log
ADGS_TEST_SEED_BASE=7000 ADGS_TEST_FACTORED_SEEDS=3 ADGS_TEST_ADVERSARIAL_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_random_configs.py -q -p no:cacheprovider -k "factored_exchange_fuzz or adversarial" > $o/fuzz2.log 2>&1; tail -3 $o/fuzz2.log
for sl in 1 0; do
  ADGS_SLABS=$sl timeout 600 python bench.py --steps 100 --warmup 20 --no-secondary --no-cpu-baseline > $o/c3_slabs$sl.json 2> $o/c3_$sl.err
  ADGS_SLABS=$sl timeout 600 python bench.py --steps 20 --warmup 5 --config C5 --no-secondary --no-cpu-baseline > $o/c5_slabs$sl.json 2> $o/c5_$sl.err
done
ADGS_BINNING=bucket timeout 600 python bench.py --steps 20 --warmup 5 --config C5 --no-secondary --no-cpu-baseline > $o/c5_bucket.json 2> $o/c5_b.err
python - <<PY
import json
for f in ("c3_slabs1","c3_slabs0","c5_slabs1","c5_slabs0","c5_bucket"):
    try:
        d = json.loads(open("$o/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k: d["stages_ms"][k] for k in ("preprocess_fwd","scan","duplicate_keys","radix_sort","tile_ranges","render_fwd")}, d["config"].get("pipeline","")[:60])
    except Exception as e:
        print(f, "failed", e)
PY
