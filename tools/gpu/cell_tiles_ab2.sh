#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/cells
for rep in 1 2; do for c in 12 11 10 9 8; do
  ADGS_CELL_TILES=$c timeout 400 python bench.py --steps 200 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null > gpurun_out/cells/c3_${c}_$rep.json
  python -c "
import sys, json
d = json.loads(open('gpurun_out/cells/c3_${c}_$rep.json').read().strip().splitlines()[-1]); c = d['config']
print('cell $c: %.1f, step %.4f ms, events %s, idle %s, stages sum %.4f, pairs %s' % (d['value'], d['ms_per_step'], c.get('step_ms_hip_events'), json.dumps(c.get('gpu_idle'))[:200], sum(d['stages_ms'].values()), c.get('cell_pairs_sorted')))"
done; done
