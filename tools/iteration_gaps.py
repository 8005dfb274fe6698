#!/usr/bin/env python3
"""Where the GPU idles inside a training iteration: from a rocprofv3 kernel trace (--kernel-trace --output-format csv) of
examples/train_iteration.py, the idle time BEFORE every kernel launch (previous kernel's end -> this kernel's start), summed per
(previous kernel -> this kernel) pair over the steady-state iterations, and the busy / idle split per iteration.

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 examples/train_iteration.py --iters 60 --json
    python tools/iteration_gaps.py out/**/*kernel_trace.csv [--marker render_fwd_v2_kernel]
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "render_fwd_v2_kernel"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < 12:
        raise SystemExit("fewer than 12 iterations in the trace")
    lo, hi = starts[len(starts) // 2], starts[-3]          # second half, without the last iterations (the run's epilogue)
    iters = sum(1 for s in starts if lo <= s < hi)
    busy = idle = 0
    gaps = collections.Counter(); counts = collections.Counter(); ktime = collections.Counter(); kcalls = collections.Counter()
    end = rows[lo][0]
    for i in range(lo, hi):
        s, e, n = rows[i]
        g = max(0, s - end)
        prev = short(rows[i - 1][2]) if i > lo else "-"
        gaps[(prev, short(n))] += g; counts[(prev, short(n))] += 1
        idle += g; busy += max(0, e - max(s, end)); end = max(end, e)
        ktime[short(n)] += e - s; kcalls[short(n)] += 1
    out = {"iterations": iters, "launches_per_iteration": round((hi - lo) / iters, 1), "busy_ms_per_iteration": round(busy / iters / 1e6, 4),
           "idle_ms_per_iteration": round(idle / iters / 1e6, 4),
           "top_gaps_us_per_iteration": [{"after": a, "before": b, "us": round(v / iters / 1e3, 2), "per_iteration": round(counts[(a, b)] / iters, 2)}
                                         for (a, b), v in gaps.most_common(40)],
           "kernels_under_10us": {"launches_per_iteration": round(sum(c for k, c in kcalls.items() if ktime[k] / c < 10e3) / iters, 1),
                                  "us_per_iteration": round(sum(t for k, t in ktime.items() if t / kcalls[k] < 10e3) / iters / 1e3, 1)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
