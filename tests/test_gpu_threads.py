"""Two host threads rendering at the same time (ctypes releases the GIL inside the library): the per-(thread, device) frame context of
api.hip -- statistics, capacity hints, mailbox, last error -- must keep them apart.  Each thread renders its own scene (different image
size, different Gaussian count) on its own stream, forward and backward, several times; every frame must equal the single-threaded result
bit for bit (images, radii) and the statistics each thread reads back must be those of its own frame."""
import threading

import pytest
import torch

from adgs import _lib, synthetic
from tests.test_gpu_raster import run_hip

pytestmark = pytest.mark.gpu


def test_two_threads_render_side_by_side():
    scenes = [synthetic.make_scene(20000, 320, 208, 260.0, seed=81, n_objects=1), synthetic.make_scene(9000, 203, 141, 180.0, seed=82, n_objects=2)]
    grads = [synthetic.make_upstream_grads(sc, 80 + i) for i, sc in enumerate(scenes)]
    ref = [run_hip(sc, grads=g) for sc, g in zip(scenes, grads)]
    ref_stats = []
    for sc, g in zip(scenes, grads):
        run_hip(sc, grads=g)
        ref_stats.append(_lib.frame_stats())
    assert ref_stats[0]["tiles"] != ref_stats[1]["tiles"] and ref_stats[0]["num_rendered"] != ref_stats[1]["num_rendered"]
    errors, barrier = [], threading.Barrier(2)

    def worker(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                barrier.wait()
                for rep in range(6):
                    out = run_hip(scenes[i], grads=grads[i])
                    st = _lib.frame_stats()                       # of THIS thread's last forward
                    assert st["tiles"] == ref_stats[i]["tiles"] and st["num_rendered"] == ref_stats[i]["num_rendered"], (i, rep, st, ref_stats[i])
                    stream.synchronize()
                    for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic", "radii"):
                        assert torch.equal(out[k], ref[i][k]), (i, rep, k)
                    for k, v in out["grads"].items():
                        if v is not None:
                            r = ref[i]["grads"][k]
                            assert torch.allclose(v, r, rtol=1e-3, atol=1e-4 * float(r.abs().max())), (i, rep, k)      # float atomics in hardware order
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
