#!/usr/bin/env python3
"""Error statistics of the HIP rasterizer against the float64 oracle, next to the float32 oracle's own error against it.

    python tools/parity_stats.py C2 [C3 ...] [--out gpurun_out/parity_stats.json]

For every output image and every gradient tensor: fraction of elements outside 1e-4, relative L2 error and the per-row
relative-error percentiles (tests/parity.py), for (HIP vs f64 oracle), (f32 oracle vs f64 oracle) and (HIP vs f32 oracle).
The float32 oracle is the reference algorithm evaluated in the reference's precision; its distance from the float64 result is the
noise floor any float32 implementation of the same algorithm has (gate flips at alpha = 1/255 included), so a HIP error within
~2x of it is what "matches the reference within its own precision" means.  ADGS_LIB selects the library build
(lib/libadgs_hip_precise.so: expf instead of v_exp_f32 in the blend kernels).
Test tooling: imports oracle/ as the checker only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from adgs import synthetic
    from tests import parity
    from tests.test_gpu_raster import run_hip, run_oracle
    res = {"lib": os.environ.get("ADGS_LIB", "default")}
    pairs = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
             ("rotations", "dL_drotations"), ("flow", "dL_dflow_points"), ("sem", "dL_dsemantic")]
    for cfg in args.configs:
        sc = synthetic.make_config_scene(cfg)
        g = synthetic.make_upstream_grads(sc, synthetic.CONFIGS[cfg]["seed"])
        t0 = time.time()
        o32 = run_oracle(sc, grads=g, precision="f32", strict=True)
        d32 = time.time() - t0
        ex = o32["explained"]
        t0 = time.time()
        h = run_hip(sc, grads=g, strict_mask=ex["pixel"])
        dh = time.time() - t0
        t0 = time.time()
        o64 = run_oracle(sc, grads=g, precision="f64")
        d64 = time.time() - t0
        print("%s: hip %.1fs, oracle f32 (two backward passes) %.1fs, oracle f64 %.1fs; radii equal: %s" % (cfg, dh, d32, d64,
              np.array_equal(h["radii"].cpu().numpy(), o32["radii"])), flush=True)
        out = {"gate_flip_mask": {"GATE_EPS": parity.GATE_EPS, "frac_pixels_flagged": ex["frac_pixel"], "frac_gaussians_fed_by_a_flagged_pixel": ex["frac_gauss"]}}
        print("  gate-flip mask (GATE_EPS %.1e): %.4f of the pixels flagged, %.4f of the Gaussians fed by a flagged pixel" % (parity.GATE_EPS, ex["frac_pixel"], ex["frac_gauss"]), flush=True)
        # the strict pass (tests/parity.py): upstream gradients zeroed at the flagged pixels on both sides -- no exemption applies
        strict = {}
        for hk, ok in pairs:
            if h["grads_strict"].get(hk) is None:
                continue
            a = h["grads_strict"][hk].cpu().numpy()
            st = parity.error_stats(a, np.asarray(o32["grads_strict"][ok]).reshape(a.shape))
            for k in ("_row_rel", "_bad", "_row_nz"):
                st.pop(k, None)
            strict["grad_" + hk] = st
            print("  " + parity.fmt_stats("strict grad_" + hk + " hip_vs_f32", st) + " n_bad=%d (allowed: 0)" % st["n_bad"], flush=True)
        out["strict_gradient_pass"] = strict
        tensors = [(k, h[k].detach().cpu().numpy(), o32[k], o64[k]) for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic")]
        tensors += [("grad_" + hk, h["grads"][hk].cpu().numpy(), o32["grads"][ok], o64["grads"][ok]) for hk, ok in pairs if h["grads"].get(hk) is not None]
        for name, a, b32, b64 in tensors:
            b32 = np.asarray(b32).reshape(a.shape); b64 = np.asarray(b64).reshape(a.shape)
            row = {}
            for tag, x, y, ex in (("hip_vs_f64", a, b64, o64["explained"]), ("f32_vs_f64", b32, b64, o64["explained"]), ("hip_vs_f32", a, b32, o32["explained"])):
                st = parity.error_stats(x, y)
                # elements outside 1e-4 that the oracle's gate margins do NOT explain (tests/parity.py: the tests allow none)
                mask = parity._broadcast_mask(ex["gauss"] if name.startswith("grad_") else ex["pixel"], x.shape, name)
                st["n_unexplained"] = int((st["_bad"] & ~mask).sum())
                st["n_flagged"] = int(mask.sum())
                for k in ("_row_rel", "_bad", "_row_nz"):
                    st.pop(k, None)
                row[tag] = st
                print("  " + parity.fmt_stats(name + " " + tag, st) + " n_bad=%d n_unexplained=%d flagged=%d" % (st["n_bad"], st["n_unexplained"], st["n_flagged"]), flush=True)
            out[name] = row
        res[cfg] = out
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
