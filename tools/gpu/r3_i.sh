#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r3i; mkdir -p $o; cd $R
run() {
  python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --cameras 1 > $o/b.json 2> $o/b.err
  python - <<PY
import json
try:
    d = json.loads(open("$o/b.json").read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("$1", d["value"], {k: round(v, 4) for k, v in s.items() if k in ("scan","duplicate_keys","radix_sort","tile_ranges","render_fwd","render_bwd","preprocess_fwd")}, d["config"].get("candidates_scanned_per_tile"), d["config"].get("cell_pairs_sorted"))
except Exception as e:
    print("$1 failed", e)
PY
}
for ct in 12 10 8 6; do ADGS_CELL_TILES=$ct run "cell_tiles=$ct"; done
ADGS_V2_PPL=2 run "ppl=2"
ADGS_V2_PPL=2 ADGS_CELL_TILES=8 run "ppl=2 cell=8"
