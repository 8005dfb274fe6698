#!/usr/bin/env python3
"""Where a wave of the forward blend kernel spends its time (experiment build: `make -C ad-gs_amd/csrc variant TAG=timing
DEFS=-DADGS_FWD_TIMING`, then `ADGS_LIB=ad-gs_amd/lib/libadgs_hip_timing.so python tools/fwd_phase_timing.py C3`): shader-clock cycles
per wave in the key-stream scan, the filter-record test, the Splat gather and the blend loop."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch

import bench
from adgs import _lib, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synthetic.CONFIGS[name]
sc = bench.build_scene(name)
cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
dev = torch.device("cuda", 0)
frame = bench.make_frame(sc, cfg, cam, dev, True)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(5):
        frame.forward()
    torch.cuda.synchronize()
    lib.adgs_test_fwd_timing(out)
    N = 20
    for _ in range(N):
        frame.forward()
    torch.cuda.synchronize()
    lib.adgs_test_fwd_timing(out)
waves = out[5]
names = ["key-stream scan (stage 1)", "filter-record test (stage 2)", "Splat gather + staging", "blend loop"]
tot = out[4] / waves
print("%s: %d waves per launch, %.0f cycles per wave" % (name, waves // N, tot))
acc = 0
for i, n in enumerate(names):
    v = out[i] / waves
    acc += v
    print("  %-32s %9.0f cycles per wave  %5.1f %%" % (n, v, 100 * v / tot))
print("  of the scan: %.1f super-rounds per wave, %.0f cycles per wave waiting for their loads (%.0f per super-round)" % (out[7] / waves, out[6] / waves, out[6] / max(out[7], 1)))
print("  of the filter test: %.1f gathers per wave, %.0f cycles waiting per gather" % (out[9] / waves, out[8] / max(out[9], 1)))
print("  %-32s %9.0f cycles per wave  %5.1f %%" % ("rest (prologue, publish, epilogue)", tot - acc, 100 * (tot - acc) / tot))
