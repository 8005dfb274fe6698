cd $GRAFT_REPO_ROOT; o=gpurun_out/graphdbg2; mkdir -p $o
python -m pytest tests/test_gpu_graph_capacity.py -m gpu -x -q -p no:cacheprovider -k "bit_for_bit" > $o/a.log 2>&1; echo "A only test2 rc=$?"
python -m pytest tests/test_gpu_graph_capacity.py -m gpu -x -q -p no:cacheprovider -k "bit_for_bit or enqueued_again" > $o/b.log 2>&1; echo "B test1+2 rc=$?"
python - > $o/c.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ad-gs_amd"))
import torch
from adgs import synthetic, graph, _lib
from tests.test_gpu_graph_capacity import _static_step
from tests.test_gpu_raster import run_hip, run_oracle
big = synthetic.make_scene(6000, 400, 300, 300.0, seed=61)
run_hip(big); run_oracle(big)          # the OpenMP oracle has run in this process
sc = synthetic.make_scene(10000, 400, 300, 300.0, sh_degree=3, seed=62, n_objects=2)
fn, leaf = _static_step(sc)
step = graph.GraphedStep(fn); step(); torch.cuda.synchronize(); print("C oracle-then-graph OK")
PY
echo "C rc=$?"; tail -3 $o/c.log
python - > $o/d.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ad-gs_amd"))
import torch
from adgs import synthetic, graph, _lib
from tests.test_gpu_graph_capacity import _static_step
from tests.test_gpu_raster import run_hip, compare
small = synthetic.make_scene(2000, 200, 136, 150.0, seed=60)
for _ in range(40): run_hip(small)
big = synthetic.make_scene(60000, 400, 300, 300.0, seed=61, scale_mult=0.02)
compare(big, grads=synthetic.make_upstream_grads(big, 61))
compare(big, grads=synthetic.make_upstream_grads(big, 61))
print("test1 body done", _lib.frame_status())
sc = synthetic.make_scene(10000, 400, 300, 300.0, sh_degree=3, seed=62, n_objects=2)
fn, leaf = _static_step(sc)
eager = [t.clone() for t in fn()]
step = graph.GraphedStep(fn); step(); torch.cuda.synchronize(); print("D test1-body-then-graph OK")
PY
echo "D rc=$?"; tail -3 $o/d.log
