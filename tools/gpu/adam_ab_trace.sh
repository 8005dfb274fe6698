#!/bin/bash
# kernel trace of tools/adam_ab.py (in-backward / separate Adam alternated in one process): per-kernel averages of each of the runs
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/adam_ab_trace; rm -rf $o; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $o/t -o t -- python3 $R/tools/adam_ab.py ${1:-2} ${2:-60} > $o/run.log 2>&1
grep "^{" $o/run.log | cut -c1-120
python3 - <<PY
import csv, glob, collections
f = glob.glob("$o/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# segments: split where the gap between consecutive render_bwd launches exceeds 50 ms (model rebuild between runs)
seg, segs, last = [], [], None
for r in rows:
    t = int(r["Start_Timestamp"])
    if last is not None and t - last > 300e6:
        segs.append(seg); seg = []
    seg.append(r); last = t
segs.append(seg)
segs = [s for s in segs if sum("render_bwd_v2" in r["Kernel_Name"] for r in s) >= 40]
names = ["render_bwd_v2", "preprocess_bwd_kernel", "deform_lin_param_grad", "render_fwd_v2", "preprocess_fwd_kernel", "sh0_rows", "adam_kernel", "deform_bwd_kernel", "l1_ssim_bwd", "envmap_bwd", "slab_sort_kernel"]
for i, s in enumerate(segs):
    acc = collections.defaultdict(list)
    for r in s[len(s) // 3:]:      # the last two thirds: steady state
        for n in names:
            if n in r["Kernel_Name"]:
                acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("run %d:" % i, {n: round(sum(v) / len(v), 1) for n, v in acc.items() if v})
PY
