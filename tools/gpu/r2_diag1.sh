#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2f; mkdir -p $o
ADGS_V2_STOP=3 timeout 300 python - > $o/chunks.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "ad-gs_amd"); sys.path.insert(0, ".")
import torch, bench
from adgs import synthetic
for cfgname in ("C3", "C5"):
    cfg = synthetic.CONFIGS[cfgname]; sc = bench.build_scene(cfgname)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    f = bench.make_frame(sc, cfg, cam, torch.device("cuda", 0), True)
    with torch.no_grad():
        f.forward()
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/$o/pmc -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$o/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R $o/pmc | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r2f/pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "chunk" in k or "cell_" in k:
            acc[k.split("(")[0][-40:]][row["Counter_Name"]] += float(row["Counter_Value"]); 
            if row["Counter_Name"] == "SQ_INSTS_VALU": cnt[k.split("(")[0][-40:]] += 1
    for k, v in acc.items():
        n = max(cnt[k], 1)
        print(k, "launches", n, {c: round(x / n) for c, x in v.items()})
PY
