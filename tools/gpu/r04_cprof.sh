#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04cprof; mkdir -p $o; cd $R
timeout 900 python - > $o/cprof.txt 2> $o/cprof.err <<'PY'
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ad-gs_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import torch
import train_iteration as ti
dev = torch.device("cuda", 0)
cfg, model, cams, env_map = ti.build("C3", 8192, dev, 16)
state = {}
off = ti.StageClock(False)
for i in range(30):
    ti.iteration(i, model, cams, env_map, off, state)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(30, 230):
    ti.iteration(i, model, cams, env_map, off, state)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(70)
print(s.getvalue())
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue())
PY
head -120 $o/cprof.txt | cut -c1-170
