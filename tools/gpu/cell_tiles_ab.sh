#!/bin/bash
# Cell edge (tiles) of the coarse binning grid: stage times and frames/s for ADGS_CELL_TILES = 12 (default at >= 4096 tiles) / 10 / 8 / 6
# gpurun -- 'bash tools/gpu/cell_tiles_ab.sh C3 100' ; 'bash tools/gpu/cell_tiles_ab.sh C5 20'
R=$GRAFT_REPO_ROOT; cd $R
cfg=${1:-C3}; steps=${2:-100}
for rep in 1 2; do for c in 12 10 8 6; do
  ADGS_CELL_TILES=$c timeout 400 python bench.py --config $cfg --steps $steps --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']; c = d['config']
print('$cfg cell $c: %.1f %s, %.4f ms | pre_fwd %.4f scan %.4f dup %.4f sort %.4f render_fwd %.4f render_bwd %.4f | pairs %s scanned/tile %s' % (d['value'], d['unit'], d['ms_per_step'], s['preprocess_fwd'], s['scan'], s['duplicate_keys'], s['radix_sort'], s['render_fwd'], s['render_bwd'], c.get('cell_pairs_sorted'), c.get('candidates_scanned_per_tile')))"
done; done
