"""bench.py's multi-rank path, end to end, on ONE GPU: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run ... bench.py --gpus 2), sharing the device, with the collectives on gloo
(ADGS_BENCH_BACKEND=gloo: RCCL refuses two ranks on one GPU).  Not a measurement -- it guards the control flow (every rank must
issue the same collectives in the same order: settle windows, warm-up, timed steps, the final barrier) and the JSON contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("config,exchange,world", [("T3", "factored", 2), ("T3", "dense", 2), ("C1", "factored", 2), ("T3", "factored", 4)])
def test_ranks_finish_and_print_one_json_line(config, exchange, world):
    env = dict(os.environ, ADGS_BENCH_BACKEND="gloo", ADGS_DP_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--config", config]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - world * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-3          # whole-job frames / max-over-ranks time
    want = "factored" if (exchange == "factored" and config == "T3") else "dense"               # static configs have no raw-SH path
    assert d["config"]["gradient_exchange"].startswith(want), d["config"]["gradient_exchange"]
    # the self-verifying multi-GPU figures: the rank count the collective library itself reports, per-rank step times and pair
    # counts, the exchange under both forms of the dense reduction
    mg = d["config"]["multi_gpu"]
    assert mg["world_size"] == world and mg["rccl_ranks"] == world
    assert len(mg["per_rank_ms_per_step"]) == world and all(x > 0 for x in mg["per_rank_ms_per_step"])
    assert len(mg["per_rank_cell_pairs_last_frame"]) == world and all(x > 0 for x in mg["per_rank_cell_pairs_last_frame"])
    assert set(mg["exchange_ms_by_collective"]) == {"all_reduce", "rs_ag"}
    assert all(v is not None and v["total"] >= 0 and v["calls"] > 0 for v in mg["exchange_ms_by_collective"].values())
    assert d["config"]["camera_pool"].startswith("16 cameras")


@pytest.mark.parametrize("world,cams,densify,collective,self_launch", [(2, 3, 0, "all_reduce", False), (4, 3, 0, "rs_ag", False), (2, 5, 3, "rs_ag", True),
                                                                        (1, 3, 3, "all_reduce", False)])
def test_iteration_mode_cameras_per_iteration_densify_and_collective_forms(world, cams, densify, collective, self_launch):
    """BASELINE.json's C4 / C5 workloads on the small T3 scene: `cams` cameras per iteration dealt round-robin over the ranks
    (4 ranks x 3 cameras: one rank idles and still takes part in every collective), optional densify/prune inside the timed loop,
    both forms of the dense reduction, and `python bench.py --gpus N` starting its own ranks."""
    env = dict(os.environ, ADGS_BENCH_BACKEND="gloo", ADGS_DP_COLLECTIVE=collective, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = ["--gpus", str(world), "--steps", "7", "--warmup", "2", "--config", "T3", "--cams-per-iter", str(cams), "--densify-every", str(densify)]
    if self_launch or world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port()), os.path.join(ROOT, "bench.py")] + tail
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 7 and d["scaling"] == "strong" and d["config"]["cameras_per_step"] == cams
    assert abs(d["value"] - cams * 7 / (d["ms_per_step"] * 7e-3)) / d["value"] < 1e-3          # cameras per second over the max-over-ranks time
    assert set(d["config"]["step_ms_hip_events"]) >= {"median", "p10", "p90"}
    if world > 1 or cams > 1:
        ex = d["config"]["exchange_ms"]
        assert ex["calls"] == 7 and ex["total"] >= 0 and {"allgather_wait", "expansion", "dense_reduction_wait"} <= set(ex)
    if densify:
        assert d["config"]["densify_ms"]["calls"] >= 7 // densify
        assert d["config"]["densify_ms"]["P_end"] > 0
    if collective == "rs_ag" and world > 1:
        assert "reduce-scatter" in d["config"]["gradient_exchange"]


@pytest.mark.parametrize("world,cams", [(2, 2), (3, 4), (3, 2), (3, 7)])
def test_multi_rank_training_loop_keeps_replicas_identical_and_matches_one_process(world, cams):
    """examples/train_dp.py (render -> losses -> backward -> factored exchange -> fused Adam -> densify/prune with a seeded sampler) as TWO
    or THREE ranks sharing the GPU (collectives on gloo) against ONE process accumulating the same cameras: every rank ends with
    bit-identical parameters, and the loss sequence of the multi-rank run follows the one-process run through the densification step.
    (3, 4): an uneven deal (2 + 1 + 1 cameras); (3, 2): one rank has NO camera in the iteration and still takes part in every collective."""
    env = dict(os.environ, ADGS_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--config", "T3", "--iters", "7", "--cams", str(cams), "--densify-every", "5"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                          str(_free_port()), os.path.join(ROOT, "examples", "train_dp.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert two.returncode == 0, two.stderr[-2000:]
    assert "replicas identical: True" in two.stdout, two.stdout[-2000:]
    env1 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_dp.py")] + args, env=env1, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert one.returncode == 0, one.stderr[-2000:]
    get = lambda out: [float(x) for x in [l for l in out.splitlines() if l.startswith("LOSSES ")][0].split()[1:]]
    a, b = get(two.stdout), get(one.stdout)
    assert len(a) == len(b) == 10
    assert all(abs(x - y) <= 2e-4 * abs(y) + 1e-7 for x, y in zip(a, b)), (a, b)
    size = lambda out: [l for l in out.splitlines() if "Gaussians at the end" in l][0].split(";")[1]
    assert size(two.stdout) == size(one.stdout)                       # the same rows were cloned / split / pruned
