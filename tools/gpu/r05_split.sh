#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r05_split; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_path.py tests/test_gpu_graph_capacity.py -q -p no:cacheprovider -x -k "not c5" > $o/tests.log 2>&1
tail -5 $o/tests.log
for sp in 1 0; do
ADGS_SPLIT_SH=$sp timeout 300 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $o/b_split$sp.json 2> $o/b_split$sp.err
python - <<PY
import json
d = json.loads(open("$o/b_split$sp.json").read().strip().splitlines()[-1]); c = d["config"]
print("split=$sp value", d["value"], c["step_ms_hip_events"], "stages", d.get("stages_ms"))
PY
done
ADGS_SPLIT_SH=1 ADGS_BENCH_GRAPH=on timeout 300 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $o/b_graph.json 2> $o/b_graph.err
python - <<PY
import json
d = json.loads(open("$o/b_graph.json").read().strip().splitlines()[-1]); c = d["config"]
print("graph split=1 value", d["value"], c["step_ms_hip_events"])
PY
