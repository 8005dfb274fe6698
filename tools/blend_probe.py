#!/usr/bin/env python3
"""Sustained shader clock and (pixel, entry) pair statistics of the two blend kernels at a BASELINE.json config.

    make -C ad-gs_amd/csrc variant TAG=probe DEFS=-DADGS_PROBE
    ADGS_LIB=ad-gs_amd/lib/libadgs_hip_probe.so python tools/blend_probe.py [C3] [frames]

The probe build makes every wave of render_fwd_v2 / render_bwd_v2 read s_memtime (shader-clock ticks) and s_memrealtime (100 MHz)
around its life and count, per entry it evaluates: the contributing (pixel, entry) pairs (gates passed), the entries with at least
one such pixel, and the active 16x4 strips.  Output: one JSON object (clock in GHz = ticks per 10 ns of real time)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from adgs import _lib, synthetic
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    device = torch.device("cuda", 0)
    lib = _lib.lib()
    read = lib.adgs_test_probe_read
    read.restype = ctypes.c_int; read.argtypes = [ctypes.c_void_p]
    cfg = synthetic.CONFIGS[config]
    sc = bench.build_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    frame = bench.make_frame(sc, cfg, cam, device, True)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [up[k].to(device) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
    buf = (ctypes.c_ulonglong * 16)()

    def step():
        torch.autograd.backward(frame.forward(), ups)
        frame.zero_grad()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    read(buf)
    for _ in range(frames):
        step()
    torch.cuda.synchronize()
    read(buf)
    v = list(buf)
    ppl = 4 if (cfg["W"] + 15) // 16 * ((cfg["H"] + 15) // 16) >= 4096 else 2
    out = {"config": config, "frames": frames, "pixels_per_lane": ppl}
    for name, o in (("render_fwd_v2", 0), ("render_bwd_v2", 8)):
        cyc, real, waves, pairs, evals, live, strips = v[o:o + 7]
        out[name] = {
            "sustained_shader_clock_GHz": round(cyc / (real * 10.0), 4) if real else None,
            "mean_wave_life_us": round(real / max(waves, 1) * 0.01, 3),
            "entries_evaluated_per_frame": evals // frames, "entries_with_a_contributing_pixel_per_frame": live // frames,
            "contributing_pixel_entry_pairs_per_frame": pairs // frames,
            "evaluated_pixel_entry_pairs_per_frame": evals * 64 * ppl // frames,
            "contributing_over_evaluated": round(pairs / max(evals * 64 * ppl, 1), 4),
            "contributing_over_pairs_of_live_entries": round(pairs / max(live * 64 * ppl, 1), 4),
            "active_strips_per_live_entry": round(strips / max(live, 1), 3),
            "shader_cycles_per_evaluated_entry_and_wave": round(cyc / max(evals, 1), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
