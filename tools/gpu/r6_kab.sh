#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of the binning kernels for several library builds: gpurun -- 'bash tools/gpu/r6_kab.sh default noslab ...'
R=$GRAFT_REPO_ROOT; L=$R/ad-gs_amd/lib
for v in "$@"; do
  lib=$L/libadgs_hip_$v.so; [ $v = default ] && lib=$L/libadgs_hip.so
  export ADGS_LIB=$lib
  bash $R/tools/gpu/r6_kstats.sh kab_$v ${CFG:+--config $CFG} | grep -E "cell_|slab_sort|tile_order|bin_prepare" | sed "s/^/$v: /" | cut -c1-150
done
