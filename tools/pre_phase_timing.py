#!/usr/bin/env python3
"""Where a workgroup of the two preprocess kernels spends its life (experiment build: preprocess.hip / preprocess_bwd.hip with
-DADGS_PRE_TIMING linked as ad-gs_amd/lib/libadgs_hip_pretiming.so; `ADGS_LIB=.../libadgs_hip_pretiming.so python tools/pre_phase_timing.py C3`):
thread 0's shader-clock cycles per workgroup in staging (issue of every load -> rows in LDS, barrier passed), compute, the wait at the
closing barrier and the output phase.  One JSON object."""
import ctypes, json, os, sys
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
fo, bo = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)()


def step():
    torch.autograd.backward(frame.forward(), ups)
    frame.zero_grad()


for _ in range(10):
    step()
torch.cuda.synchronize()
lib.adgs_test_pre_timing(fo, 1); lib.adgs_test_preb_timing(bo, 1)
for _ in range(N):
    step()
torch.cuda.synchronize()
lib.adgs_test_pre_timing(fo, 0); lib.adgs_test_preb_timing(bo, 0)
f, b = list(fo), list(bo)
res = {"config": name, "frames": N}
n = max(f[5], 1)
res["preprocess_fwd"] = {"workgroups_per_launch": f[5] // N, "cycles_per_workgroup": round(f[4] / n),
                         **{k: {"cycles": round(f[i] / n), "share": round(f[i] / max(f[4], 1), 4)} for i, k in enumerate(("staging", "compute_and_stores", "closing_barrier_wait", "counts_row_out"))}}
n = max(b[13], 1)
res["preprocess_bwd"] = {"workgroups_per_launch": b[13] // N, "cycles_per_workgroup": round(b[12] / n),
                         **{k: {"cycles": round(b[8 + i] / n), "share": round(b[8 + i] / max(b[12], 1), 4)} for i, k in enumerate(("staging", "compute", "closing_barrier_wait", "gradient_rows_out"))}}
print(json.dumps(res, indent=1))
