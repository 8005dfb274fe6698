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
        ms0 = torch.cuda.memory_stats()
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
        ms1 = torch.cuda.memory_stats()
        # device allocations (hipMalloc / hipFree by the caching allocator) inside the timed loop: a loop that keeps asking the driver for memory stalls
        alloc = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("segment.all.allocated", "segment.all.freed", "num_alloc_retries", "num_device_alloc", "num_device_free")}
        print(json.dumps({"adam_in_backward": mode, "ms": round(ms, 4), "backward": s["backward"], "adam": s["adam_gaussians"], "host_backward": clock.host_summary().get("backward") if hasattr(clock, "host_summary") else None,
                          "driver_allocations_in_loop": alloc, "reserved_GiB": round(torch.cuda.memory_reserved() / 2**30, 2), "p_m_v_MiB": where}), flush=True)
        del model, cams, env_map, state
        torch.cuda.empty_cache()
