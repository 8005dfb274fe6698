#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_e; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_raster.py tests/test_gpu_train_step.py -q -p no:cacheprovider -rf > $o/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  +Assert" $o/tests.log | cut -c1-300 | tail -12
ADGS_TEST_SEED_BASE=7125 ADGS_TEST_SEEDS=1 ADGS_TEST_VARIANT_SEEDS=1 ADGS_TEST_LARGE_SEEDS=1 ADGS_TEST_ADVERSARIAL_SEEDS=1 timeout 300 python -m pytest tests/test_gpu_random_configs.py -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python tools/parity_stats.py C2 C3 --out $o/parity_stats.json > $o/parity_stats.txt 2>&1
grep -E "gate-flip|strict|^C[23]" $o/parity_stats.txt | cut -c1-260
for ct in 8 10 12 14; do
ADGS_CELL_TILES=$ct timeout 300 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $o/ct$ct.json 2> $o/ct$ct.err
python - <<PY
import json
d = json.loads(open("$o/ct$ct.json").read().strip().splitlines()[-1]); c = d["config"]; st = d.get("stages_ms") or {}
print("cell_tiles $ct value %.1f median %.4f fwd %.4f bwd %.4f binning %.4f" % (d["value"], c["step_ms_hip_events"]["median"], st.get("render_fwd", 0), st.get("render_bwd", 0), sum(st.get(k, 0) for k in ("scan", "duplicate_keys", "radix_sort", "tile_ranges"))))
PY
done
