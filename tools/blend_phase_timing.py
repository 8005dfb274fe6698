#!/usr/bin/env python3
"""Where the waves of the two blend kernels spend their time (experiment build: `make -C ad-gs_amd/csrc variant TAG=timing
DEFS=-DADGS_PHASE_TIMING`, then `ADGS_LIB=ad-gs_amd/lib/libadgs_hip_timing.so python tools/blend_phase_timing.py C3`): shader-clock
cycles per wave in the forward's key-stream scan, Splat gather + tile test, batch set-up (barrier, pool block draw) and blend loop, and in the backward's chunk-header
wait, id + Splat gather, entry loop (of it: reduction + atomic).  One JSON object."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch

import bench
from adgs import _lib, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synthetic.CONFIGS[name]
sc = bench.build_scene(name)
cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
dev = torch.device("cuda", 0)
frame = bench.make_frame(sc, cfg, cam, dev, True)
up = synthetic.make_upstream_grads(sc, 0)
ups = [up[k].to(dev) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 32)()


def step():
    torch.autograd.backward(frame.forward(), ups)
    frame.zero_grad()


for _ in range(10):
    step()
torch.cuda.synchronize()
lib.adgs_test_phase_timing(out)
for _ in range(N):
    step()
torch.cuda.synchronize()
lib.adgs_test_phase_timing(out)
v = list(out)
res = {"config": name, "frames": N}
waves = max(v[5], 1)
tot = v[4] / waves
f = {"waves_per_launch": v[5] // N, "cycles_per_wave": round(tot)}
for i, n in enumerate(["key_stream_scan", "splat_gather_and_tile_test", "batch_setup", "blend_loop"]):
    f[n] = {"cycles_per_wave": round(v[i] / waves), "share": round(v[i] / waves / tot, 4)}
f["rest_prologue_publish_epilogue_share"] = round(1.0 - sum(v[0:4]) / waves / tot, 4)
f["scan_super_rounds_per_wave"] = round(v[7] / waves, 2)
f["cycles_waiting_per_scan_super_round"] = round(v[6] / max(v[7], 1))
f["gather_rounds_per_wave"] = round(v[9] / waves, 2)
f["cycles_waiting_per_gather_round"] = round(v[8] / max(v[9], 1))
res["render_fwd_v2"] = f
b = v[16:32]
waves = max(b[5], 1)
tot = b[4] / waves
g = {"waves_per_launch": b[5] // N, "cycles_per_wave": round(tot), "chunks_per_wave": round(b[6] / waves, 2), "entries_per_wave": round(b[7] / waves, 1)}
for i, n in ((8, "prologue_pixel_state_loads"), (0, "chunk_header_wait"), (1, "id_and_splat_gather"), (2, "entry_loop")):
    g[n] = {"cycles_per_wave": round(b[i] / waves), "share": round(b[i] / waves / tot, 4)}
g["entry_loop"]["cycles_per_entry"] = round(b[2] / max(b[7], 1), 1)
g["reduction_and_atomic"] = {"cycles_per_entry": round(b[3] / max(b[7], 1), 1), "share_of_wave": round(b[3] / waves / tot, 4)}
g["cycles_per_chunk_header_wait"] = round(b[0] / max(b[6], 1))
g["cycles_per_chunk_gather"] = round(b[1] / max(b[6], 1))
res["render_bwd_v2"] = g
print(json.dumps(res, indent=1))
