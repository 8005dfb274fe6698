#!/bin/bash
# frames/s at C3 (and C5 iteration mode) against the coarse-cell edge, bucket binning
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/${1:-cells}; mkdir -p $o
for ct in 12 10 8 6; do
  ADGS_CELL_TILES=$ct python bench.py --no-cpu-baseline --no-secondary > $o/c3_$ct.json 2> $o/c3_$ct.err
  python - $o/c3_$ct.json $ct <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("C3 cell", sys.argv[2], d["value"], d["config"]["step_ms_hip_events"]["median"], {k:round(v,3) for k,v in d["stages_ms"].items()}, d["config"].get("cell_pairs_sorted"))
PY
done
for ct in 12 8 6; do
  ADGS_CELL_TILES=$ct python bench.py --config C5 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > $o/c5_$ct.json 2> $o/c5_$ct.err
  python - $o/c5_$ct.json $ct <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("C5 cell", sys.argv[2], d["value"], d["config"]["pipeline"][:60], d["config"].get("cell_pairs_sorted"))
PY
done
