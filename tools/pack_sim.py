#!/usr/bin/env python3
"""What a better TILE ORDER could buy the two blend kernels, from measured wave lives (experiment build of tools/wave_timeline.py).

The hardware hands workgroups out in blockIdx order to whichever slot frees first -- list scheduling with the tile order as the list.
Today's list is "longest first" (tile_order_kernel: consumed entries, descending).  With ~2.3 jobs per slot that is the bad regime of
LPT: the shortest jobs start last, on slots that already carry two long ones.  This script replays the measured lives through a list
scheduler (calibration: today's order must reproduce the measured kernel time) and then through other lists: "banded" (the slots that
will take one job more than the others are given short jobs only, from the start), with the true lives and with lives PREDICTED from
the consumed counts (what a kernel could compute).

    collect (GPU):  ADGS_LIB=ad-gs_amd/lib/libadgs_hip_timeline.so python tools/pack_sim.py collect C3 gpurun_out/pack_sim_c3.npz
    simulate (CPU): python tools/pack_sim.py sim gpurun_out/pack_sim_c3.npz
"""
import heapq, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(config, path):
    for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
        sys.path.insert(0, p)
    import ctypes, torch
    import bench
    from adgs import _lib, synthetic, deform
    from diff_gaussian_rasterization import _C
    device = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[config]
    sc = bench.build_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    frame = bench.make_frame(sc, cfg, cam, device, True)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [up[k].to(device) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
    lib = _lib.lib()
    lib.adgs_test_v2_tile_words.restype = ctypes.c_longlong
    lib.adgs_test_v2_tile_words.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    out = {}
    for which in ("fwd", "bwd"):
        os.environ["ADGS_TIMELINE_BWD"] = "1" if which == "bwd" else "0"
        for _ in range(5):
            torch.autograd.backward(frame.forward(), ups); frame.zero_grad()
        with torch.no_grad():
            pkg = deform.get_deformed_pkg(frame.model, frame.t) if hasattr(frame, "model") else None
        s = bench.make_settings(cfg, cam, sc, device)
        e = torch.empty(0, device=device)
        if pkg is not None:
            t = dict(means3D=pkg["xyz"], opacities=pkg["opacity"], scales=pkg["scales"], rotations=pkg["rotation"], shs=pkg["shs"])
            flow = frame.model.get_deformed_xyz(frame.t + 0.05)
        else:
            t = {k: v.detach() for k, v in frame.leaf.items()}; flow = frame.flow
        sem = frame.sem
        leafs = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
        for rep in range(3):      # the third render of these camera tensors: its forward runs in the order the second one left
            r = _C.rasterize_gaussians(s.bg, leafs["means3D"], e, leafs["opacities"], leafs["scales"], leafs["rotations"], s.scale_modifier, e, s.viewmatrix, s.projmatrix,
                                       s.tanfovx, s.tanfovy, s.image_height, s.image_width, leafs["shs"], flow, sem, s.sh_degree, s.campos, s.prefiltered, s.inv_depth, False)
            _C.rasterize_gaussians_backward(s.bg, leafs["means3D"], r[4], e, leafs["scales"], leafs["rotations"], s.scale_modifier, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                            s.tanfovy, ups[0], ups[1], ups[3], ups[4], sem, flow, leafs["shs"], s.sh_degree, s.campos, r[5], r[0], r[6], r[7], r[3], ups[2],
                                            s.inv_depth, False)
            torch.cuda.synchronize()
            W, H = int(s.image_width), int(s.image_height)
            n = int(lib.adgs_test_v2_tile_counters(r[7].data_ptr(), W, H, None, None, 0, None))
            a = (ctypes.c_uint32 * n)(); b = (ctypes.c_uint32 * n)(); c = (ctypes.c_uint32 * n)()
            lib.adgs_test_v2_tile_words(r[7].data_ptr(), W, H, a, b, n, None)
            lib.adgs_test_v2_tile_counters(r[7].data_ptr(), W, H, c, None, n, None)
            tag = which if rep == 2 else "%s_prev%d" % (which, 2 - rep)      # fwd_prev1: the render before the one simulated
            out[tag + "_start"] = np.frombuffer(a, np.uint32).copy(); out[tag + "_end"] = np.frombuffer(b, np.uint32).copy()
            out[tag + "_consumed"] = np.frombuffer(c, np.uint32).copy()
    np.savez_compressed(path, **out)
    print("wrote", path)


def lives(start, end):
    start = start.astype(np.int64); end = end.astype(np.int64)
    end = np.where(end < start, end + (1 << 32), end)
    t0 = start.min()
    return (start - t0) * 0.01, (end - start) * 0.01, float((end - t0).max() * 0.01)


def list_schedule(order, life, slots, queues=1):
    """Jobs in `order` to the earliest free slot; queues > 1: job i belongs to queue i % queues (the XCD round-robin), slots / queues each."""
    heaps = [[0.0] * (slots // queues) for _ in range(queues)]
    end = 0.0
    for i, j in enumerate(order):
        h = heaps[i % queues]
        t = heapq.heappop(h) + life[j]
        heapq.heappush(h, t)
        end = max(end, t)
    return end


def banded_order(pred, slots):
    """Slots that take m + 1 jobs get the SHORTEST (m + 1) * rem jobs, the others the longest m each; inside a class the jobs of a slot are
    drawn from bands of the sorted class, alternating direction (long + short ...); the list = all jobs by their intended start time."""
    n = len(pred)
    by = np.argsort(-pred, kind="stable")          # longest first
    m, rem = divmod(n, slots)
    plan = []                                       # (start time, job)
    def fill(jobs, per_slot):                       # jobs: longest first, len = per_slot * k
        k = len(jobs) // per_slot
        if k == 0: return
        bands = [jobs[b * k:(b + 1) * k] for b in range(per_slot)]
        for s in range(k):
            t = 0.0
            for b in range(per_slot):
                j = bands[b][s] if b % 2 == 0 else bands[b][k - 1 - s]
                plan.append((t, int(j))); t += pred[j]
    n_long = m * (slots - rem)
    fill(by[:n_long], m)
    fill(by[n_long:], m + 1)
    plan.sort(key=lambda x: x[0])
    return np.array([j for _, j in plan])


def sim(path, slots_by_kernel=None):
    d = np.load(path)
    res = {}
    for which in ("fwd", "bwd"):
        start, life, kernel_us = lives(d[which + "_start"], d[which + "_end"])
        cons = d[which + "_consumed"].astype(np.float64)
        n = len(life)
        t = np.sort(np.concatenate([start, start + life])); alive_max = 0
        ev = sorted([(s, 1) for s in start] + [(s + l, -1) for s, l in zip(start, life)]); a = 0
        for _, x in ev:
            a += x; alive_max = max(alive_max, a)
        slots = alive_max
        actual = np.argsort(start, kind="stable")              # the order the hardware started them in
        A = np.polyfit(cons, life, 1); pred = np.polyval(A, cons)
        prev = lives(d[which + "_prev1_start"], d[which + "_prev1_end"])[1] if which + "_prev1_start" in d else None
        r = {"tiles": n, "measured_kernel_us": round(kernel_us, 1), "slots(max in flight)": int(slots), "sum_lives/slots_us": round(float(life.sum() / slots), 1),
             "longest_life_us": round(float(life.max()), 1), "life~consumed": {"fit_us": [round(float(x), 4) for x in A], "corr": round(float(np.corrcoef(cons, life)[0, 1]), 3),
             "rms_residual_us": round(float(np.std(life - pred)), 1)}}
        for q in (1, 8):
            key = "sim_us[%d queue%s]" % (q, "s" if q > 1 else "")
            r[key] = {"as_started": round(list_schedule(actual, life, slots, q), 1),
                      "longest_first_by_consumed": round(list_schedule(np.argsort(-cons, kind="stable"), life, slots, q), 1),
                      "longest_first_by_true_life": round(list_schedule(np.argsort(-life, kind="stable"), life, slots, q), 1),
                      "banded_by_true_life": round(list_schedule(banded_order(life, slots), life, slots, q), 1),
                      "banded_by_predicted_life": round(list_schedule(banded_order(pred, slots), life, slots, q), 1),
                      "shortest_first": round(list_schedule(np.argsort(cons, kind="stable"), life, slots, q), 1),
                      "row_major": round(list_schedule(np.arange(n), life, slots, q), 1)}
            if prev is not None:      # the lives the same tiles had in the PREVIOUS render of the camera as the prediction
                r[key]["longest_first_by_previous_life"] = round(list_schedule(np.argsort(-prev, kind="stable"), life, slots, q), 1)
                r[key]["banded_by_previous_life"] = round(list_schedule(banded_order(prev, slots), life, slots, q), 1)
        if prev is not None:
            r["life~previous_life"] = {"corr": round(float(np.corrcoef(prev, life)[0, 1]), 3), "rms_difference_us": round(float(np.std(life - prev)), 1)}
        res["render_%s_v2" % which] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "collect": collect(sys.argv[2], sys.argv[3])
    else: sim(sys.argv[2])
