"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
Units (MI355X_MICROARCH.md, HBM section): both counters are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
so wide coalesced reads are doubled.  Calibrated here on known launches of the same run: at::FillFunctor<float> writing
the 12 MB means2D tensor reports WRITE_SIZE 11.7 MiB (x1), sort_scatter reading 12 B/pair reports FETCH_SIZE/2."""
import csv, sys, collections, json, re, hashlib, os
def lib_stamp():      # the build the counters were collected on (bench.py flags committed counters of another build as stale)
    path = os.environ.get("ADGS_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ad-gs_amd", "lib", "libadgs_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None
def agg(path):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(float); c = collections.Counter()
    for r in rows:
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"].split("(adgs")[0].split("(int")[0].split("(unsigned")[0])
        k = k.replace("void ", "").strip()
        a[k] += float(r["Counter_Value"]); c[k] += 1
    return {k: a[k] / c[k] for k in a}, c
fetch, cf = agg(sys.argv[1]); write, cw = agg(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    rd = 2.0 * fetch.get(k, 0.0) * 1024; wr = write.get(k, 0.0) * 1024
    out[k] = {"launches_sampled": int(cf.get(k, cw.get(k, 0))), "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr)}
json.dump(dict(out, _library_sha256_16=lib_stamp()), open(sys.argv[3], "w"), indent=1)
if len(sys.argv) > 4:
    # per-frame total: all launches of the run (bench.py with ADGS_BENCH_SKIP_STATS=1) divided by the number of frames
    # = launches of the blend backward (exactly one per frame)
    # (each pass by ITS OWN frame count: the two passes are two runs, and until round 6 both were divided by the fetch pass's count --
    # a first r06 profile whose passes differed in length (217 / 117 frames) reported 1.91 GB where the kernels sum to 2.33)
    bwd = lambda c: max(c.get(next((k for k in c if "render_bwd_v2_kernel" in k), ""), 1), 1)
    frames, frames_w = bwd(cf), bwd(cw)
    tot = sum(2.0 * fetch.get(k, 0.0) * cf.get(k, 0) * 1024 for k in fetch) / frames + sum(write.get(k, 0.0) * cw.get(k, 0) * 1024 for k in write) / frames_w
    json.dump({"frames": int(frames), "frames_write_pass": int(frames_w), "hbm_bytes_per_frame": int(tot), "_library_sha256_16": lib_stamp(),
               "note": "sum over ALL kernels of a run of 2*FETCH_SIZE (KiB) / its frames + WRITE_SIZE (KiB) / its frames"}, open(sys.argv[4], "w"), indent=1)
    print("frames", frames, frames_w, "HBM bytes per frame %.1f MB" % (tot / 1e6))
for k, v in list(out.items())[:24]:
    print(f"{k[:70]:70s} rd {v['read_bytes_per_launch']/1e6:9.1f} MB  wr {v['write_bytes_per_launch']/1e6:9.1f} MB")
