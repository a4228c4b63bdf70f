"""one rank, exchange forced: the data-parallel step variants with and without the recorded step plan (diagnostic).
usage: python scripts/ab_dp_plan.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1


def timed(fn, eng, n):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) * 1e6 / n
    eng.flush_deferred()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, t_host


for name, make in (("all-reduce", lambda e: DataParallelStep(e, force_exchange=True)), ("sharded", lambda e: ShardedDataParallelStep(e, force_exchange=True))):
    for plan in (True, False):
        for chain in (False, True):
            eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
            step = make(eng)
            eng.use_plan = plan
            eng.chain_priority = chain
            t, h = timed(lambda: step.train_step(x), eng, iters)
            print("%-10s plan %-5s chain stream %-5s  %8.1f us per step   host %7.1f us" % (name, plan, chain, t, h), flush=True)
            del step, eng
