"""SQ / GRBM counters of the two blend kernels from rocprofv3 --pmc passes of bench.py:
    python tools/pmc_blend.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
Any number of passes (one CSV each; a counter may appear in one of them); per-launch averages per kernel.

Normalisation (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count QUAD-cycles summed over all
waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  What the figures mean:
  * waves_per_simd_mean = 4 SQ_WAVE_CYCLES / (SIMDs x kernel cycles): resident waves, averaged over the kernel's life;
  * wave_state_shares: SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (parked at s_waitcnt / barrier) and SQ_WAIT_INST_ANY (ready, not issued) over
    SQ_WAVE_CYCLES -- disjoint, they add up to ~1;
  * active_cycles_per_valu_inst = 4 SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: how long a wave sits in the "executing a VALU instruction"
    state per instruction (~4.2: the instruction's trip through the pipe).  Round 3 divided the SAME counter by SIMD cycles and called
    it "VALU busy" -- 0.98 / 1.11: above 1 because the trips of different waves overlap in the pipeline (a full-rate instruction
    occupies the issue port for ~2.3 cycles but its wave for ~4.2).  It is a per-WAVE state, not a pipe utilisation, and is not
    reported as one any more;
  * valu_insts_per_simd_cycle = SQ_INSTS_VALU / (SIMDs x kernel cycles): the issue rate; the port's ceiling depends on the mix
    (tools/microbench/issue_hazards.hip: 1 / 2.3 full-rate fp32, 1 / 4.2 .. 4.4 compares / selects / min / DPP, 1 / 8.2 exp / rcp;
    the kernels' own mixes: tools/isa_mix.py).
"""
import collections, csv, json, sys, hashlib, os

SIMDS = 1024.0


def lib_stamp():
    path = os.environ.get("ADGS_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ad-gs_amd", "lib", "libadgs_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def agg(paths):
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for path in paths:
        for r in csv.DictReader(open(path)):
            for k in ("render_fwd_v2_kernel", "render_bwd_v2_kernel"):
                if k in r["Kernel_Name"]:
                    a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in a[k].items()} for k in a}


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    c = agg(paths)
    out = {"note": "rocprofv3 --pmc, per-launch averages; the normalisation of every figure is described in tools/pmc_blend.py", "kernels": {}}
    for k, v in c.items():
        d = {name: round(val, 2) for name, val in sorted(v.items())}
        if "GRBM_GUI_ACTIVE" in v:
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            d["kernel_cycles"] = round(cyc, 1)
            if "SQ_WAVE_CYCLES" in v:
                d["waves_per_simd_mean"] = round(4.0 * v["SQ_WAVE_CYCLES"] / (SIMDS * cyc), 3)
            if "SQ_INSTS_VALU" in v:
                d["valu_insts_per_simd_cycle"] = round(v["SQ_INSTS_VALU"] / SIMDS / cyc, 4)
            if "SQ_INSTS_SALU" in v:
                d["salu_insts_per_simd_cycle"] = round(v["SQ_INSTS_SALU"] / SIMDS / cyc, 4)
            if "SQ_INSTS_LDS" in v:
                d["lds_insts_per_cu_cycle"] = round(v["SQ_INSTS_LDS"] / (SIMDS / 4.0) / cyc, 4)
            for name in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"):
                if name in v:
                    d[name.lower() + "_per_cu_cycle"] = round(v[name] / (SIMDS / 4.0) / cyc, 4)
        if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            w = v["SQ_WAVE_CYCLES"]
            d["wave_state_shares"] = {n.lower()[3:]: round(v[n] / w, 4) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                                                                                 "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                                                                                 "SQ_ACTIVE_INST_MISC") if n in v}
        if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_INSTS_VALU"):
            d["active_cycles_per_valu_inst"] = round(4.0 * v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"], 3)
        if "SQ_WAVES" in v:
            d["waves_launched"] = round(v["SQ_WAVES"], 1)
        out["kernels"][k] = d
    out["_library_sha256_16"] = lib_stamp()
    json.dump(out, open(out_path, "w"), indent=1)
    for k, d in out["kernels"].items():
        print(k, {x: d[x] for x in ("kernel_cycles", "waves_per_simd_mean", "valu_insts_per_simd_cycle", "active_cycles_per_valu_inst", "wave_state_shares") if x in d})


if __name__ == "__main__":
    main()
