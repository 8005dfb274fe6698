"""profiles/r01/pmc_blend_kernels.json from two rocprofv3 --pmc passes of bench.py (SQ counters; GRBM_GUI_ACTIVE):
    python tools/pmc_blend.py <sq counter_collection.csv> <grbm counter_collection.csv> <out.json>
Per-launch averages of the two blend kernels; valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel cycles) with
kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (SQ *_CYCLES / ACTIVE / WAIT counters are quad-cycles summed over all waves)."""
import collections, csv, json, sys, hashlib, os
def lib_stamp():
    path = os.environ.get("ADGS_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ad-gs_amd", "lib", "libadgs_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None
def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(path)):
        for k in ("render_fwd_v2_kernel", "render_bwd_v2_kernel"):
            if k in r["Kernel_Name"]:
                a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in a[k].items()} for k in a}
sq, gr = agg(sys.argv[1]), agg(sys.argv[2])
out = {"note": "rocprofv3 --pmc, per launch averages, bench.py C3 defaults; SQ_* *_CYCLES/ACTIVE/WAIT counters are quad-cycles summed over all waves, "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, PMC section)", "kernels": {}}
for k in sq:
    d = {c: round(v, 4) for c, v in sq[k].items()}
    d["GRBM_GUI_ACTIVE"] = round(gr[k]["GRBM_GUI_ACTIVE"], 1)
    d["kernel_cycles"] = round(gr[k]["GRBM_GUI_ACTIVE"] / 8.0, 3)
    raw = 4.0 * sq[k]["SQ_ACTIVE_INST_VALU"] / (1024.0 * d["kernel_cycles"])
    d["valu_active_raw"] = round(raw, 4)            # can exceed 1: the counter sums the waves in flight in a SIMD's VALU pipeline
    d["valu_busy_frac"] = round(min(raw, 1.0), 4)
    d["valu_insts_per_simd"] = round(sq[k]["SQ_INSTS_VALU"] / 1024.0, 4)
    d["valu_insts_per_simd_cycle"] = round(sq[k]["SQ_INSTS_VALU"] / 1024.0 / d["kernel_cycles"], 4)     # full-rate fp32 peak: 0.5 (a wave64 op issues in 2 cycles)
    out["kernels"][k] = d
out["_library_sha256_16"] = lib_stamp()
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, "VALU busy %.3f" % d["valu_busy_frac"], "cycles %.0f" % d["kernel_cycles"], "VALU insts/SIMD %.0f" % d["valu_insts_per_simd"])
