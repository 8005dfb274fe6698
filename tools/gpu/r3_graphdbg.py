"""Debug: which variant of a captured static frame crashes in capture_end?  Each case runs in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASE = r'''
import sys, os
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/ad-gs_amd")
import torch
from adgs import synthetic, graph, _lib
from tests.test_gpu_graph_capacity import _static_step
from tests.test_gpu_raster import run_hip
case = "%(case)s"
P, W, H, f = 10000, 400, 300, 300.0
kw = dict(sh_degree=3, seed=62, n_objects=2)
if "c1like" in case:
    kw = dict(sh_degree=0, seed=0)
if "prehint" in case:
    big = synthetic.make_scene(60000, 400, 300, 300.0, seed=61, scale_mult=0.02)
    run_hip(big)
sc = synthetic.make_scene(P, W, H, f, **kw)
fn, leaf = _static_step(sc)
if "gradsonly" in case:
    full = fn
    fn = lambda: full()[4:]
if "eagerfirst" in case:
    fn(); torch.cuda.synchronize()
step = graph.GraphedStep(fn)
step(); torch.cuda.synchronize()
print(case, "OK", _lib.frame_status())
'''
for case in ("plain", "gradsonly", "eagerfirst", "c1like", "prehint", "prehint_gradsonly"):
    r = subprocess.run([sys.executable, "-c", CASE % dict(root=ROOT, case=case)], capture_output=True, text=True, timeout=600)
    print("=== %s rc=%d" % (case, r.returncode))
    print(r.stdout[-400:])
    if r.returncode != 0:
        print(r.stderr[-1500:])
