#!/usr/bin/env python3
"""A/B of the training iteration with and without the in-backward Adam step, alternated in ONE process (and so on one box, one
allocator history): python tools/adam_ab.py [rounds] [iters].  Prints, per run, ms per iteration, the backward / Adam stage times and
where the allocator put the parameter and its two moments (address differences of p, m, v of the two largest tensors, MiB)."""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("train_iteration", os.path.join(ROOT, "examples", "train_iteration.py"))
ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
import torch
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
for r in range(rounds):
    for mode in (True, False):
        cfg, model, cams, env_map = ti.build("C3", 8192, dev, 16, mode)
        state, off = {}, ti.StageClock(False)
        for i in range(12):
            ti.iteration(i, model, cams, env_map, off, state)
        torch.cuda.synchronize()
        clock = ti.StageClock(True)
        t0 = time.perf_counter()
        for i in range(12, 12 + iters):
            ti.iteration(i, model, cams, env_map, clock, state)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / iters
        where = {}
        for name in ("_scene_shs_rest", "shs_deform_param_scene"):
            p = getattr(model, name); st = model.optimizer.state[p]
            where[name] = [round((st["exp_avg"].data_ptr() - p.data_ptr()) / 2**20, 3), round((st["exp_avg_sq"].data_ptr() - p.data_ptr()) / 2**20, 3),
                           hex(p.data_ptr() & 0xfffff)]
        s = clock.summary()
        print(json.dumps({"adam_in_backward": mode, "ms": round(ms, 4), "backward": s["backward"], "adam": s["adam_gaussians"], "p_m_v_MiB": where}), flush=True)
        del model, cams, env_map, state
        torch.cuda.empty_cache()
