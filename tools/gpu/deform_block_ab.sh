#!/bin/bash
# Stage times of the deformation kernels for workgroup sizes 256 / 128 / 64 (variant library with the ADGS_DEFORM_BLOCK knob): alternated twice
# gpurun -- 'bash tools/gpu/deform_block_ab.sh'
R=$GRAFT_REPO_ROOT; cd $R
export ADGS_LIB=$R/ad-gs_amd/lib/libadgs_hip_dblk.so
for rep in 1 2; do for b in 256 128 64; do
  ADGS_DEFORM_BLOCK=$b timeout 300 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('block $b: %.1f frames/s, deform_fwd %.4f deform_bwd %.4f preprocess_fwd %.4f' % (d['value'], s['deform_fwd'], s['deform_bwd'], s['preprocess_fwd']))"
done; done
