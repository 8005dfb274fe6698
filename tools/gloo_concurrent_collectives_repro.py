"""Stand-alone reproducer of a gloo behaviour the multi-rank dry runs ran into (nothing of this repository is imported):
with FOUR or more ranks, an asynchronous all-gather of device tensors hangs when all-reduces are issued while it is still in
flight ("flat" / "list": hang inside w.wait(); "seq": the gather is waited for first and everything completes).  RCCL executes a
communicator's collectives in issue order and is not affected; adgs.dp therefore serialises the gather only when the group's
backend is gloo.

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/gloo_concurrent_collectives_repro.py flat|list|seq
"""
import os, sys, faulthandler, torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
mode = sys.argv[1]
n = 60000 + 6 * 4000
send = torch.randn(1, n, device="cuda"); recv = torch.empty(world, 1, n, device="cuda")
a = torch.randn(54016, device="cuda"); b = torch.randn(300, device="cuda")
for it in range(400):
    if mode in ("flat", "seq"):
        w = dist.all_gather_into_tensor(recv.view(-1), send[:1].reshape(-1), async_op=True)
    else:
        outs = [recv[r].view(-1) for r in range(world)]
        w = dist.all_gather(outs, send[:1].reshape(-1), async_op=True)
    if mode == "seq": w.wait()
    w1 = dist.all_reduce(a, async_op=True); w2 = dist.all_reduce(b, async_op=True)
    w.wait(); w1.wait(); w2.wait()
    a.mul_(0.25); b.mul_(0.25)
torch.cuda.synchronize()
if rank == 0: print("repro", mode, "ok")
dist.destroy_process_group()
