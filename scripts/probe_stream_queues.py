"""Which HIP streams share a hardware queue, and what it costs.  The HIP runtime multiplexes a process's streams
onto GPU_MAX_HW_QUEUES (4) hardware queues; streams on one queue block each other.  One rank, RCCL group of size 1, exchange forced:
a pool of K torch streams is created and TOUCHED in order (a stream gets its queue at first use), then the sharded data-parallel step
and the plain step are timed with (side stream, communication stream) = (pool[i], pool[j]) for a grid of i, j.
usage: python scripts/probe_stream_queues.py [K] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
warm = torch.zeros(1024, device=dev)
dist.all_reduce(warm)                        # the collective library takes its streams NOW
torch.cuda.synchronize()
pool = [torch.cuda.Stream(device=dev) for _ in range(K)]
for s in pool:                               # first use in a fixed order
    with torch.cuda.stream(s):
        warm.add_(1.0)
torch.cuda.synchronize()
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1


def timed(fn, eng, n):
    for _ in range(12):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    eng.flush_deferred()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"# GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default 4)')}  pool of {K} streams touched in order; us per step")
print("# plain step (caller = default stream, side = pool[i])")
row = []
for i in range(K):
    eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
    eng._side = pool[i]
    row.append(timed(lambda: eng.train_step(x), eng, iters))
    del eng
print("side=pool[i]: " + " ".join(f"{v:7.0f}" for v in row))
for name, cls in (("sharded", ShardedDataParallelStep), ("all-reduce", DataParallelStep)):
    print(f"# {name} step, rows side = pool[i], columns comm = pool[j]")
    for i in range(K):
        row = []
        for j in range(K):
            if i == j:
                row.append(float("nan")); continue
            eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
            eng._side = pool[i]
            st = cls(eng, force_exchange=True)
            if cls is ShardedDataParallelStep:
                st.comm_stream = pool[j]
            else:
                st.reducer.comm_stream = pool[j]
            row.append(timed(lambda: st.train_step(x), eng, iters))
            del st, eng
        print(f"side=pool[{i}]: " + " ".join(f"{v:7.0f}" for v in row))
dist.destroy_process_group()
